// Rows.java -- live row arrays of a librec DenseMatrix (row(i,false) is a shallow view, SURVEY 8b).
// Source only: NOT compiled or tested here.
package carskit.alg.gpu;

import librec.data.DenseMatrix;

final class Rows {
    static double[][] of(DenseMatrix m) {
        double[][] rows = new double[m.numRows()][];
        for (int i = 0; i < rows.length; i++) rows[i] = m.row(i, false).getData();
        return rows;
    }
}
