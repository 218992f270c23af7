// Dev.java -- where a model container goes: the single instance behind a cmi_handle, or the sharded group behind a
// cmi_group_handle (-Dcarskit.shards=N).  GpuSupport hands the *_GPU classes a group as the NEGATED handle (native addresses are
// positive), so that their copyIn / copyOut stay one line per container.  No JDK here: not compiled by javac; executed under the Java-source interpreter (tests/test_java_binding_exec.py).
package carskit.alg.gpu;

final class Dev {
    private Dev() {}

    static long ofGroup(long g) { return -g; }
    static boolean isGroup(long h) { return h < 0; }

    static void setMatrix(long h, int which, double[][] rows) {
        if (h < 0) NativeMF.groupSetMatrix(-h, which, rows);
        else NativeMF.setMatrix(h, which, rows);
    }
    static void getMatrix(long h, int which, double[][] rows) {
        if (h < 0) NativeMF.groupGetMatrix(-h, which, rows);
        else NativeMF.getMatrix(h, which, rows);
    }
    static void setVector(long h, int which, double[] v) {
        if (h < 0) NativeMF.groupSetVector(-h, which, v);
        else NativeMF.setVector(h, which, v);
    }
    static void getVector(long h, int which, double[] v) {
        if (h < 0) NativeMF.groupGetVector(-h, which, v);
        else NativeMF.getVector(h, which, v);
    }
}
