// Rows.java -- live row arrays of a librec DenseMatrix (row(i,false) is a shallow view, SURVEY 8b), and the dense image of the
// guava Table<Integer,Integer,Double> CAMF_CUCI keeps its context-bias tables in (CAMF_CUCI.java:43-66: every (row, condition)
// cell is present).  No JDK here: not compiled by javac; executed under the Java-source interpreter (tests/test_java_binding_exec.py).
package carskit.alg.gpu;

import com.google.common.collect.Table;
import librec.data.DenseMatrix;
import librec.data.SymmMatrix;

final class Rows {
    private Rows() {}

    static double[][] of(DenseMatrix m) {
        double[][] rows = new double[m.numRows()][];
        for (int i = 0; i < rows.length; i++) rows[i] = m.row(i, false).getData();
        return rows;
    }

    static double[][] ofTable(Table<Integer, Integer, Double> t, int numRows, int numCols) {
        double[][] rows = new double[numRows][numCols];
        for (int i = 0; i < numRows; i++)
            for (int c = 0; c < numCols; c++) {
                Double v = t.get(i, c);
                rows[i][c] = v == null ? 0.0 : v;
            }
        return rows;
    }

    /** ccMatrix_ICS (librec SymmMatrix, CAMF.java:45) as the full symmetric dense image the C ABI exchanges */
    static double[][] ofSymm(SymmMatrix m, int dim) {
        double[][] rows = new double[dim][dim];
        for (int i = 0; i < dim; i++)
            for (int j = 0; j < dim; j++) rows[i][j] = m.get(i, j);
        return rows;
    }

    static void intoSymm(SymmMatrix m, double[][] rows) {
        for (int i = 0; i < rows.length; i++)
            for (int j = i; j < rows.length; j++) m.set(i, j, rows[i][j]);
    }

    /** cmi_get_state into a fresh dense image (for containers that are not double[][] on the Java side) */
    static double[][] fetch(long h, int which, int numRows, int numCols) {
        double[][] rows = new double[numRows][numCols];
        Dev.getMatrix(h, which, rows);
        return rows;
    }

    static void intoTable(Table<Integer, Integer, Double> t, double[][] rows) {
        for (int i = 0; i < rows.length; i++)
            for (int c = 0; c < rows[i].length; c++) t.put(i, c, rows[i][c]);
    }
}
