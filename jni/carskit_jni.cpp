// carskit_jni.cpp -- JNI shim: carskit.alg.gpu.NativeMF -> include/carskit_mi355x.h.  No logic lives here.
// Build (only where a JDK exists; NOT built or tested in this image, which has no jni.h):
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/carskit_jni.cpp \
//       -Lcarskit_amd/lib -lcarskit_mi355x -o libcarskit_mi355x_jni.so
#include <jni.h>

#include <vector>

#include "carskit_mi355x.h"

static void throw_cmi(JNIEnv *env, cmi_handle h, int rc) {
    if (rc == CMI_OK) return;
    env->ThrowNew(env->FindClass("java/lang/RuntimeException"), cmi_last_error(h));
}

extern "C" {

JNIEXPORT jlong JNICALL Java_carskit_alg_gpu_NativeMF_create(JNIEnv *env, jclass, jint model, jint k, jint nu,
                                                            jint ni, jint nc, jint device, jint flags) {
    cmi_handle h = nullptr;
    int rc = cmi_create(model, k, nu, ni, nc, device, (unsigned)flags, &h);
    if (rc != CMI_OK) throw_cmi(env, nullptr, rc);
    return (jlong)h;
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_destroy(JNIEnv *, jclass, jlong h) { cmi_destroy((cmi_handle)h); }

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setRatingsCsr(JNIEnv *env, jclass, jlong h, jintArray rowPtr,
                                                                   jintArray colInd, jdoubleArray data,
                                                                   jintArray uiUser, jintArray uiItem,
                                                                   jintArray ctxPtr, jintArray ctxConds) {
    const jsize nRows = env->GetArrayLength(rowPtr) - 1, n = env->GetArrayLength(colInd);
    jint *rp = (jint *)env->GetPrimitiveArrayCritical(rowPtr, nullptr);
    jint *uu = (jint *)env->GetPrimitiveArrayCritical(uiUser, nullptr);
    jint *ui = (jint *)env->GetPrimitiveArrayCritical(uiItem, nullptr);
    std::vector<int32_t> u((size_t)n), j((size_t)n); // expand CSR rows to per-tuple (user, item)
    for (jsize r = 0; r < nRows; ++r)
        for (jint q = rp[r]; q < rp[r + 1]; ++q) {
            u[(size_t)q] = uu[r];
            j[(size_t)q] = ui[r];
        }
    env->ReleasePrimitiveArrayCritical(uiItem, ui, JNI_ABORT);
    env->ReleasePrimitiveArrayCritical(uiUser, uu, JNI_ABORT);
    env->ReleasePrimitiveArrayCritical(rowPtr, rp, JNI_ABORT);
    jint *ci = (jint *)env->GetPrimitiveArrayCritical(colInd, nullptr);
    jdouble *d = (jdouble *)env->GetPrimitiveArrayCritical(data, nullptr);
    jint *cp = (jint *)env->GetPrimitiveArrayCritical(ctxPtr, nullptr);
    jint *cc = (jint *)env->GetPrimitiveArrayCritical(ctxConds, nullptr);
    int rc = cmi_set_ratings((cmi_handle)h, n, u.data(), j.data(), (const int32_t *)ci, d,
                             env->GetArrayLength(ctxPtr) - 1, (const int32_t *)cp, (const int32_t *)cc);
    env->ReleasePrimitiveArrayCritical(ctxConds, cc, JNI_ABORT);
    env->ReleasePrimitiveArrayCritical(ctxPtr, cp, JNI_ABORT);
    env->ReleasePrimitiveArrayCritical(data, d, JNI_ABORT);
    env->ReleasePrimitiveArrayCritical(colInd, ci, JNI_ABORT);
    throw_cmi(env, (cmi_handle)h, rc);
}

static void matrix_io(JNIEnv *env, jlong h, jint which, jobjectArray rows, bool set) {
    const jsize nr = env->GetArrayLength(rows);
    if (nr == 0) return;
    jdoubleArray r0 = (jdoubleArray)env->GetObjectArrayElement(rows, 0);
    const jsize nc = env->GetArrayLength(r0);
    std::vector<double> flat((size_t)nr * nc);
    if (!set) {
        int rc = cmi_get_state((cmi_handle)h, which, flat.data(), (int64_t)flat.size(), CMI_DTYPE_F64);
        if (rc != CMI_OK) return throw_cmi(env, (cmi_handle)h, rc);
    }
    for (jsize i = 0; i < nr; ++i) {
        jdoubleArray row = (jdoubleArray)env->GetObjectArrayElement(rows, i);
        if (set) env->GetDoubleArrayRegion(row, 0, nc, flat.data() + (size_t)i * nc);
        else env->SetDoubleArrayRegion(row, 0, nc, flat.data() + (size_t)i * nc);
        env->DeleteLocalRef(row);
    }
    if (set) throw_cmi(env, (cmi_handle)h, cmi_set_state((cmi_handle)h, which, flat.data(), (int64_t)flat.size(), CMI_DTYPE_F64));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setMatrix(JNIEnv *env, jclass, jlong h, jint w, jobjectArray rows) { matrix_io(env, h, w, rows, true); }
JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_getMatrix(JNIEnv *env, jclass, jlong h, jint w, jobjectArray rows) { matrix_io(env, h, w, rows, false); }

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setVector(JNIEnv *env, jclass, jlong h, jint w, jdoubleArray v) {
    jdouble *p = env->GetDoubleArrayElements(v, nullptr);
    int rc = cmi_set_state((cmi_handle)h, w, p, env->GetArrayLength(v), CMI_DTYPE_F64);
    env->ReleaseDoubleArrayElements(v, p, JNI_ABORT);
    throw_cmi(env, (cmi_handle)h, rc);
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_getVector(JNIEnv *env, jclass, jlong h, jint w, jdoubleArray v) {
    jdouble *p = env->GetDoubleArrayElements(v, nullptr);
    int rc = cmi_get_state((cmi_handle)h, w, p, env->GetArrayLength(v), CMI_DTYPE_F64);
    env->ReleaseDoubleArrayElements(v, p, 0);
    throw_cmi(env, (cmi_handle)h, rc);
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setHparams(JNIEnv *env, jclass, jlong h, jdouble ru, jdouble ri,
                                                                jdouble rb, jdouble rc_, jdouble gm) {
    throw_cmi(env, (cmi_handle)h, cmi_set_hparams((cmi_handle)h, ru, ri, rb, rc_, gm));
}

JNIEXPORT jdouble JNICALL Java_carskit_alg_gpu_NativeMF_trainEpoch(JNIEnv *env, jclass, jlong h, jdouble lr) {
    double loss = 0;
    throw_cmi(env, (cmi_handle)h, cmi_train_epoch((cmi_handle)h, lr, &loss));
    return loss;
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_evalRatings(JNIEnv *env, jclass, jlong h, jintArray u,
                                                                         jintArray j, jintArray ctx, jdoubleArray r,
                                                                         jdouble lo, jdouble hi) {
    const jsize n = env->GetArrayLength(u);
    jint *pu = env->GetIntArrayElements(u, nullptr), *pj = env->GetIntArrayElements(j, nullptr);
    jint *pc = ctx ? env->GetIntArrayElements(ctx, nullptr) : nullptr;
    jdouble *pr = env->GetDoubleArrayElements(r, nullptr);
    double out[6];
    int64_t cnt = 0;
    int rc = cmi_eval_ratings((cmi_handle)h, n, (const int32_t *)pu, (const int32_t *)pj, (const int32_t *)pc, pr, lo, hi, out, &cnt);
    out[5] = (double)cnt;
    env->ReleaseDoubleArrayElements(r, pr, JNI_ABORT);
    if (pc) env->ReleaseIntArrayElements(ctx, pc, JNI_ABORT);
    env->ReleaseIntArrayElements(j, pj, JNI_ABORT);
    env->ReleaseIntArrayElements(u, pu, JNI_ABORT);
    throw_cmi(env, (cmi_handle)h, rc);
    jdoubleArray res = env->NewDoubleArray(6);
    env->SetDoubleArrayRegion(res, 0, 6, out);
    return res;
}

} // extern "C"
