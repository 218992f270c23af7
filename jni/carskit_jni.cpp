// carskit_jni.cpp -- JNI shim: carskit.alg.gpu.NativeMF -> include/carskit_mi355x.h.  No logic lives here: every function
// copies its Java arrays into native buffers, makes ONE C-ABI call (or the get/set pair of one container) and turns a non-zero
// status into a RuntimeException carrying cmi_last_error().
// Build (only where a JDK exists; NOT built or run in this image, which has no jni.h):
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/carskit_jni.cpp
//       -Lcarskit_amd/lib -lcarskit_mi355x -o libcarskit_mi355x_jni.so          (one command line)
// tests/test_java_binding_text.py keeps this file and java/carskit/alg/gpu/NativeMF.java in step (names and JNI signatures).
//
// Arrays are COPIED out with Get<Type>ArrayRegion before any library call: cmi_set_ratings builds its schedule for seconds and
// blocks on HIP, which must not happen inside a Get/ReleasePrimitiveArrayCritical region (the JVM may stall every other thread's
// GC for that long, and no other JNI call is legal inside one).
#include <jni.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "carskit_mi355x.h"

namespace {

void throw_msg(JNIEnv *env, const char *msg) { env->ThrowNew(env->FindClass("java/lang/RuntimeException"), msg); }
void throw_cmi(JNIEnv *env, cmi_handle h, int rc) {
    if (rc != CMI_OK) throw_msg(env, cmi_last_error(h));
}
void throw_fm(JNIEnv *env, cmi_fm_handle h, int rc) {
    if (rc != CMI_OK) throw_msg(env, cmi_fm_last_error(h));
}
void throw_group(JNIEnv *env, cmi_group_handle g, int rc) {
    if (rc != CMI_OK) throw_msg(env, cmi_group_last_error(g));
}
// Mismatched Java arrays must become an exception, never an out-of-bounds access in native memory: every entry point
// checks lengths (and CSR row pointers) before the one C-ABI call.  Returns true when it has thrown.
bool bad_args(JNIEnv *env, bool bad, const char *msg) {
    if (bad) env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), msg);
    return bad;
}

std::vector<int32_t> ints(JNIEnv *env, jintArray a) {
    std::vector<int32_t> v;
    if (!a) return v;
    v.resize((std::size_t)env->GetArrayLength(a));
    if (!v.empty()) env->GetIntArrayRegion(a, 0, (jsize)v.size(), reinterpret_cast<jint *>(v.data()));
    return v;
}
std::vector<double> doubles(JNIEnv *env, jdoubleArray a) {
    std::vector<double> v;
    if (!a) return v;
    v.resize((std::size_t)env->GetArrayLength(a));
    if (!v.empty()) env->GetDoubleArrayRegion(a, 0, (jsize)v.size(), v.data());
    return v;
}
jdoubleArray to_java(JNIEnv *env, const double *p, std::size_t n) {
    jdoubleArray res = env->NewDoubleArray((jsize)n);
    if (res && n) env->SetDoubleArrayRegion(res, 0, (jsize)n, p);
    return res;
}
// rows of a double[][] <-> one row-major buffer
std::vector<double> flatten(JNIEnv *env, jobjectArray rows, jsize *nr_out, jsize *nc_out) {
    const jsize nr = rows ? env->GetArrayLength(rows) : 0;
    jsize nc = 0;
    if (nr > 0) {
        jdoubleArray r0 = (jdoubleArray)env->GetObjectArrayElement(rows, 0);
        nc = env->GetArrayLength(r0);
        env->DeleteLocalRef(r0);
    }
    std::vector<double> flat((std::size_t)nr * (std::size_t)nc);
    for (jsize i = 0; i < nr; ++i) {
        jdoubleArray row = (jdoubleArray)env->GetObjectArrayElement(rows, i);
        env->GetDoubleArrayRegion(row, 0, nc, flat.data() + (std::size_t)i * nc);
        env->DeleteLocalRef(row);
    }
    *nr_out = nr;
    *nc_out = nc;
    return flat;
}
void scatter(JNIEnv *env, jobjectArray rows, const std::vector<double> &flat, jsize nr, jsize nc) {
    for (jsize i = 0; i < nr; ++i) {
        jdoubleArray row = (jdoubleArray)env->GetObjectArrayElement(rows, i);
        env->SetDoubleArrayRegion(row, 0, nc, flat.data() + (std::size_t)i * nc);
        env->DeleteLocalRef(row);
    }
}
// rowPtr starts at 0, never decreases and ends at the number of stored entries; data and (if given) the pair maps cover it
bool bad_csr(JNIEnv *env, const std::vector<int32_t> &row_ptr, std::size_t nnz, std::size_t n_data, const std::vector<int32_t> *ui_user,
             const std::vector<int32_t> *ui_item) {
    bool bad = row_ptr.empty() || row_ptr.front() != 0 || (std::size_t)row_ptr.back() != nnz || n_data != nnz;
    for (std::size_t r = 0; !bad && r + 1 < row_ptr.size(); ++r) bad = row_ptr[r + 1] < row_ptr[r];
    if (!bad && ui_user) bad = ui_user->size() + 1 < row_ptr.size() || ui_item->size() + 1 < row_ptr.size();
    return bad_args(env, bad, "CSR arrays inconsistent: rowPtr must start at 0, be non-decreasing and end at colInd.length == data.length; "
                              "uiUser / uiItem must cover every row");
}
// tuple arrays of one call: same length (ctx / r may be absent)
bool bad_tuples(JNIEnv *env, std::size_t n, std::size_t nj, bool has_ctx, std::size_t nc, bool has_r, std::size_t nr) {
    return bad_args(env, nj != n || (has_ctx && nc != n) || (has_r && nr != n), "tuple arrays (u, j, ctx, r) differ in length");
}
// CSR rows (user-item pair ids) -> per-tuple (user, item): what `for (MatrixEntry me : trainMatrix)` yields
void expand_pairs(const std::vector<int32_t> &row_ptr, const std::vector<int32_t> &ui_user, const std::vector<int32_t> &ui_item,
                  std::size_t n, std::vector<int32_t> &u, std::vector<int32_t> &j) {
    u.assign(n, 0);
    j.assign(n, 0);
    for (std::size_t r = 0; r + 1 < row_ptr.size(); ++r)
        for (int32_t q = row_ptr[r]; q < row_ptr[r + 1]; ++q) {
            u[(std::size_t)q] = ui_user[r];
            j[(std::size_t)q] = ui_item[r];
        }
}

} // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_carskit_alg_gpu_NativeMF_create(JNIEnv *env, jclass, jint model, jint k, jint nUsers, jint nItems,
                                                            jint nConds, jint device, jint flags) {
    cmi_handle h = nullptr;
    const int rc = cmi_create(model, k, nUsers, nItems, nConds, device, (unsigned)flags, &h);
    if (rc != CMI_OK) throw_cmi(env, nullptr, rc);
    return (jlong)h;
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_destroy(JNIEnv *, jclass, jlong h) { cmi_destroy((cmi_handle)h); }

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setRatingsCsr(JNIEnv *env, jclass, jlong h, jintArray rowPtr, jintArray colInd,
                                                                   jdoubleArray data, jintArray uiUser, jintArray uiItem,
                                                                   jintArray ctxPtr, jintArray ctxConds) {
    const std::vector<int32_t> rp = ints(env, rowPtr), ci = ints(env, colInd), uu = ints(env, uiUser), ui = ints(env, uiItem),
                               cp = ints(env, ctxPtr), cc = ints(env, ctxConds);
    const std::vector<double> d = doubles(env, data);
    if (bad_csr(env, rp, ci.size(), d.size(), &uu, &ui)) return;
    if (bad_args(env, cp.empty() || cp.front() != 0 || (std::size_t)cp.back() != cc.size(), "ctxPtr must start at 0 and end at ctxConds.length")) return;
    std::vector<int32_t> u, j;
    expand_pairs(rp, uu, ui, ci.size(), u, j);
    throw_cmi(env, (cmi_handle)h,
              cmi_set_ratings((cmi_handle)h, (int64_t)ci.size(), u.data(), j.data(), ci.data(), d.data(), (int32_t)cp.size() - 1, cp.data(),
                              cc.data()));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setRatings2D(JNIEnv *env, jclass, jlong h, jintArray rowPtr, jintArray colInd,
                                                                  jdoubleArray data) {
    const std::vector<int32_t> rp = ints(env, rowPtr), ci = ints(env, colInd);
    const std::vector<double> d = doubles(env, data);
    if (bad_csr(env, rp, ci.size(), d.size(), nullptr, nullptr)) return;
    std::vector<int32_t> u(ci.size());
    for (std::size_t r = 0; r + 1 < rp.size(); ++r)
        for (int32_t q = rp[r]; q < rp[r + 1]; ++q) u[(std::size_t)q] = (int32_t)r; // the 2-D train matrix: row = user, column = item
    throw_cmi(env, (cmi_handle)h, cmi_set_ratings((cmi_handle)h, (int64_t)ci.size(), u.data(), ci.data(), nullptr, d.data(), 0, nullptr, nullptr));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setMatrix(JNIEnv *env, jclass, jlong h, jint which, jobjectArray rows) {
    jsize nr = 0, nc = 0;
    const std::vector<double> flat = flatten(env, rows, &nr, &nc);
    throw_cmi(env, (cmi_handle)h, cmi_set_state((cmi_handle)h, which, flat.data(), (int64_t)flat.size(), CMI_DTYPE_F64));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_getMatrix(JNIEnv *env, jclass, jlong h, jint which, jobjectArray rows) {
    jsize nr = 0, nc = 0;
    std::vector<double> flat = flatten(env, rows, &nr, &nc); // for the shape
    const int rc = cmi_get_state((cmi_handle)h, which, flat.data(), (int64_t)flat.size(), CMI_DTYPE_F64);
    if (rc != CMI_OK) return throw_cmi(env, (cmi_handle)h, rc);
    scatter(env, rows, flat, nr, nc);
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setVector(JNIEnv *env, jclass, jlong h, jint which, jdoubleArray v) {
    const std::vector<double> p = doubles(env, v);
    throw_cmi(env, (cmi_handle)h, cmi_set_state((cmi_handle)h, which, p.data(), (int64_t)p.size(), CMI_DTYPE_F64));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_getVector(JNIEnv *env, jclass, jlong h, jint which, jdoubleArray v) {
    std::vector<double> p((std::size_t)env->GetArrayLength(v));
    const int rc = cmi_get_state((cmi_handle)h, which, p.data(), (int64_t)p.size(), CMI_DTYPE_F64);
    if (rc != CMI_OK) return throw_cmi(env, (cmi_handle)h, rc);
    env->SetDoubleArrayRegion(v, 0, (jsize)p.size(), p.data());
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setHparams(JNIEnv *env, jclass, jlong h, jdouble regU, jdouble regI, jdouble regB,
                                                                jdouble regC, jdouble globalMean) {
    throw_cmi(env, (cmi_handle)h, cmi_set_hparams((cmi_handle)h, regU, regI, regB, regC, globalMean));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setDeviceShare(JNIEnv *env, jclass, jlong h, jint instances) {
    throw_cmi(env, (cmi_handle)h, cmi_set_device_share((cmi_handle)h, (int)instances));
}

JNIEXPORT jdouble JNICALL Java_carskit_alg_gpu_NativeMF_trainEpoch(JNIEnv *env, jclass, jlong h, jdouble lRate) {
    double loss = 0;
    throw_cmi(env, (cmi_handle)h, cmi_train_epoch((cmi_handle)h, lRate, &loss));
    return loss;
}

JNIEXPORT jint JNICALL Java_carskit_alg_gpu_NativeMF_train(JNIEnv *env, jclass, jlong h, jint numIters, jdouble initLRate,
                                                           jdouble maxLRate, jint boldDriver, jdouble decay, jint earlyStop,
                                                           jdoubleArray losses, jdoubleArray lrates) {
    std::vector<double> ls((std::size_t)(numIters > 0 ? numIters : 0)), rs(ls.size());
    int run = 0;
    const int rc = cmi_train((cmi_handle)h, numIters, initLRate, maxLRate, boldDriver, decay, earlyStop, ls.data(), rs.data(), &run, nullptr);
    if (losses && run > 0) env->SetDoubleArrayRegion(losses, 0, run, ls.data());
    if (lrates && run > 0) env->SetDoubleArrayRegion(lrates, 0, run, rs.data());
    throw_cmi(env, (cmi_handle)h, rc);
    return run;
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_evalRatings(JNIEnv *env, jclass, jlong h, jintArray u, jintArray j,
                                                                         jintArray ctx, jdoubleArray r, jdouble minRate,
                                                                         jdouble maxRate) {
    const std::vector<int32_t> pu = ints(env, u), pj = ints(env, j), pc = ints(env, ctx);
    const std::vector<double> pr = doubles(env, r);
    if (bad_tuples(env, pu.size(), pj.size(), ctx != nullptr, pc.size(), true, pr.size())) return nullptr;
    double out[6] = {0, 0, 0, 0, 0, 0};
    int64_t cnt = 0;
    const int rc = cmi_eval_ratings((cmi_handle)h, (int64_t)pu.size(), pu.data(), pj.data(), ctx ? pc.data() : nullptr, pr.data(), minRate,
                                    maxRate, out, &cnt);
    out[5] = (double)cnt;
    if (rc != CMI_OK) {
        throw_cmi(env, (cmi_handle)h, rc);
        return nullptr;
    }
    return to_java(env, out, 6);
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setEvalRatings(JNIEnv *env, jclass, jlong h, jintArray u, jintArray j,
                                                                    jintArray ctx, jdoubleArray r) {
    const std::vector<int32_t> pu = ints(env, u), pj = ints(env, j), pc = ints(env, ctx);
    const std::vector<double> pr = doubles(env, r);
    if (bad_tuples(env, pu.size(), pj.size(), ctx != nullptr, pc.size(), true, pr.size())) return;
    throw_cmi(env, (cmi_handle)h,
              cmi_set_eval_ratings((cmi_handle)h, (int64_t)pu.size(), pu.data(), pj.data(), ctx ? pc.data() : nullptr, pr.data()));
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_evalResident(JNIEnv *env, jclass, jlong h, jdouble minRate,
                                                                          jdouble maxRate) {
    double out[6] = {0, 0, 0, 0, 0, 0};
    int64_t cnt = 0;
    const int rc = cmi_eval_resident((cmi_handle)h, minRate, maxRate, out, &cnt);
    out[5] = (double)cnt;
    if (rc != CMI_OK) {
        throw_cmi(env, (cmi_handle)h, rc);
        return nullptr;
    }
    return to_java(env, out, 6);
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_predictBatch(JNIEnv *env, jclass, jlong h, jintArray u, jintArray j,
                                                                          jintArray ctx, jint bound, jdouble lo, jdouble hi) {
    const std::vector<int32_t> pu = ints(env, u), pj = ints(env, j), pc = ints(env, ctx);
    if (bad_tuples(env, pu.size(), pj.size(), ctx != nullptr, pc.size(), false, 0)) return nullptr;
    std::vector<double> out(pu.size());
    const int rc = cmi_predict_batch((cmi_handle)h, (int64_t)pu.size(), pu.data(), pj.data(), ctx ? pc.data() : nullptr, bound, lo, hi, out.data());
    if (rc != CMI_OK) {
        throw_cmi(env, (cmi_handle)h, rc);
        return nullptr;
    }
    return to_java(env, out.data(), out.size());
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_evalRankings(JNIEnv *env, jclass, jlong h, jintArray tu, jintArray tj,
                                                                          jintArray tctx, jdoubleArray tr, jintArray su, jintArray sj,
                                                                          jintArray sctx, jdoubleArray sr, jdouble binThold,
                                                                          jint numRecs, jint numIgnore, jint strategy) {
    const std::vector<int32_t> a = ints(env, tu), b = ints(env, tj), c = ints(env, tctx), d = ints(env, su), e = ints(env, sj), f = ints(env, sctx);
    const std::vector<double> ra = doubles(env, tr), rb = doubles(env, sr);
    if (bad_tuples(env, a.size(), b.size(), tctx != nullptr, c.size(), true, ra.size()) ||
        bad_tuples(env, d.size(), e.size(), sctx != nullptr, f.size(), true, rb.size()))
        return nullptr;
    double out[CMI_RANK_MEASURES];
    int64_t nq = 0;
    const int rc = cmi_eval_rankings((cmi_handle)h, (int64_t)a.size(), a.data(), b.data(), tctx ? c.data() : nullptr, ra.data(), (int64_t)d.size(),
                                     d.data(), e.data(), sctx ? f.data() : nullptr, rb.data(), binThold, numRecs, numIgnore, strategy, out, &nq,
                                     nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc != CMI_OK) {
        throw_cmi(env, (cmi_handle)h, rc);
        return nullptr;
    }
    return to_java(env, out, CMI_RANK_MEASURES);
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_saveModel(JNIEnv *env, jclass, jlong h, jstring path, jdouble lRate, jdouble lastLoss,
                                                               jint epochsDone) {
    const char *p = env->GetStringUTFChars(path, nullptr);
    const int rc = cmi_save_model((cmi_handle)h, p, lRate, lastLoss, epochsDone);
    env->ReleaseStringUTFChars(path, p);
    throw_cmi(env, (cmi_handle)h, rc);
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_loadModel(JNIEnv *env, jclass, jlong h, jstring path) {
    const char *p = env->GetStringUTFChars(path, nullptr);
    double out[3] = {0, 0, 0};
    int done = 0;
    const int rc = cmi_load_model((cmi_handle)h, p, &out[0], &out[1], &done);
    env->ReleaseStringUTFChars(path, p);
    out[2] = (double)done;
    if (rc != CMI_OK) {
        throw_cmi(env, (cmi_handle)h, rc);
        return nullptr;
    }
    return to_java(env, out, 3);
}

// ---- FM -----------------------------------------------------------------------------------------------------------

JNIEXPORT jlong JNICALL Java_carskit_alg_gpu_NativeMF_fmCreate(JNIEnv *env, jclass, jint k, jint nUsers, jint nItems, jint nConds,
                                                              jint nCtxDims, jint device, jint flags) {
    cmi_fm_handle h = nullptr;
    const int rc = cmi_fm_create(k, nUsers, nItems, nConds, nCtxDims, device, (unsigned)flags, &h);
    if (rc != CMI_OK) throw_fm(env, nullptr, rc);
    return (jlong)h;
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_fmDestroy(JNIEnv *, jclass, jlong h) { cmi_fm_destroy((cmi_fm_handle)h); }

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_fmSetHparams(JNIEnv *env, jclass, jlong h, jdouble regLw, jdouble regLf,
                                                                  jlong globalSize) {
    throw_fm(env, (cmi_fm_handle)h, cmi_fm_set_hparams((cmi_fm_handle)h, regLw, regLf, (int64_t)globalSize));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_fmSetRatingsCsr(JNIEnv *env, jclass, jlong h, jintArray rowPtr, jintArray colInd,
                                                                     jdoubleArray data, jintArray uiUser, jintArray uiItem) {
    const std::vector<int32_t> rp = ints(env, rowPtr), ci = ints(env, colInd), uu = ints(env, uiUser), ui = ints(env, uiItem);
    const std::vector<double> d = doubles(env, data);
    if (bad_csr(env, rp, ci.size(), d.size(), &uu, &ui)) return;
    std::vector<int32_t> u, j;
    expand_pairs(rp, uu, ui, ci.size(), u, j);
    throw_fm(env, (cmi_fm_handle)h, cmi_fm_set_ratings((cmi_fm_handle)h, (int64_t)ci.size(), u.data(), j.data(), ci.data(), d.data()));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_fmSetModel(JNIEnv *env, jclass, jlong h, jdouble w0, jdoubleArray w,
                                                                jobjectArray vRows) {
    const std::vector<double> pw = doubles(env, w);
    jsize nr = 0, nc = 0;
    const std::vector<double> flat = flatten(env, vRows, &nr, &nc);
    throw_fm(env, (cmi_fm_handle)h, cmi_fm_set_model((cmi_fm_handle)h, w0, pw.data(), flat.data()));
}

JNIEXPORT jdouble JNICALL Java_carskit_alg_gpu_NativeMF_fmGetModel(JNIEnv *env, jclass, jlong h, jdoubleArray w, jobjectArray vRows) {
    std::vector<double> pw((std::size_t)env->GetArrayLength(w));
    jsize nr = 0, nc = 0;
    std::vector<double> flat = flatten(env, vRows, &nr, &nc); // for the shape
    double w0 = 0;
    const int rc = cmi_fm_get_model((cmi_fm_handle)h, &w0, pw.data(), flat.data());
    if (rc != CMI_OK) {
        throw_fm(env, (cmi_fm_handle)h, rc);
        return 0;
    }
    env->SetDoubleArrayRegion(w, 0, (jsize)pw.size(), pw.data());
    scatter(env, vRows, flat, nr, nc);
    return w0;
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_fmTrain(JNIEnv *env, jclass, jlong h, jint numIters) {
    throw_fm(env, (cmi_fm_handle)h, cmi_fm_train((cmi_fm_handle)h, numIters));
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_fmPredictBatch(JNIEnv *env, jclass, jlong h, jintArray u, jintArray j,
                                                                            jintArray ctx, jint bound, jdouble lo, jdouble hi) {
    const std::vector<int32_t> pu = ints(env, u), pj = ints(env, j), pc = ints(env, ctx);
    if (bad_tuples(env, pu.size(), pj.size(), true, pc.size(), false, 0)) return nullptr;
    std::vector<double> out(pu.size());
    const int rc = cmi_fm_predict_batch((cmi_fm_handle)h, (int64_t)pu.size(), pu.data(), pj.data(), pc.data(), bound, lo, hi, out.data());
    if (rc != CMI_OK) {
        throw_fm(env, (cmi_fm_handle)h, rc);
        return nullptr;
    }
    return to_java(env, out.data(), out.size());
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_fmEvalRankings(JNIEnv *env, jclass, jlong h, jintArray tu, jintArray tj,
                                                                            jintArray tctx, jdoubleArray tr, jintArray su, jintArray sj,
                                                                            jintArray sctx, jdoubleArray sr, jdouble binThold,
                                                                            jint numRecs, jint numIgnore, jint strategy) {
    const std::vector<int32_t> a = ints(env, tu), b = ints(env, tj), c = ints(env, tctx), d = ints(env, su), e = ints(env, sj), f = ints(env, sctx);
    const std::vector<double> ra = doubles(env, tr), rb = doubles(env, sr);
    if (bad_tuples(env, a.size(), b.size(), true, c.size(), true, ra.size()) || bad_tuples(env, d.size(), e.size(), true, f.size(), true, rb.size()))
        return nullptr;
    double out[CMI_RANK_MEASURES];
    int64_t nq = 0;
    const int rc = cmi_fm_eval_rankings((cmi_fm_handle)h, (int64_t)a.size(), a.data(), b.data(), c.data(), ra.data(), (int64_t)d.size(), d.data(),
                                        e.data(), f.data(), rb.data(), binThold, numRecs, numIgnore, strategy, out, &nq, nullptr, nullptr,
                                        nullptr, nullptr, nullptr);
    if (rc != CMI_OK) {
        throw_fm(env, (cmi_fm_handle)h, rc);
        return nullptr;
    }
    return to_java(env, out, CMI_RANK_MEASURES);
}

// ---- CAMF_ICS / LCS / MCS: EmptyContextConditions, numF, numContextDims (cmi_set_sim_params) --------------------------

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_setSimParams(JNIEnv *env, jclass, jlong h, jint numF, jint nCtxDims,
                                                                  jintArray emptyConds) {
    const std::vector<int32_t> e = ints(env, emptyConds);
    throw_cmi(env, (cmi_handle)h, cmi_set_sim_params((cmi_handle)h, numF, nCtxDims, e.data(), (int)e.size()));
}

// ---- one recommender sharded over several GPUs (cmi_group_*; -Dcarskit.shards=N) ------------------------------------------

JNIEXPORT jlong JNICALL Java_carskit_alg_gpu_NativeMF_groupCreate(JNIEnv *env, jclass, jint model, jint k, jint nUsers, jint nItems,
                                                                 jint nConds, jint nShards, jintArray devices, jint flags) {
    const std::vector<int32_t> dev = ints(env, devices);
    if (bad_args(env, devices != nullptr && dev.size() != (std::size_t)nShards, "devices must be null or hold one index per shard")) return 0;
    cmi_group_handle g = nullptr;
    const int rc = cmi_group_create(model, k, nUsers, nItems, nConds, nShards, devices ? reinterpret_cast<const int *>(dev.data()) : nullptr,
                                    (unsigned)flags, &g);
    if (rc != CMI_OK) throw_group(env, nullptr, rc);
    return (jlong)g;
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupDestroy(JNIEnv *, jclass, jlong g) { cmi_group_destroy((cmi_group_handle)g); }

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupSetHparams(JNIEnv *env, jclass, jlong g, jdouble regU, jdouble regI, jdouble regB,
                                                                     jdouble regC, jdouble globalMean) {
    throw_group(env, (cmi_group_handle)g, cmi_group_set_hparams((cmi_group_handle)g, regU, regI, regB, regC, globalMean));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupSetRatingsCsr(JNIEnv *env, jclass, jlong g, jintArray rowPtr, jintArray colInd,
                                                                        jdoubleArray data, jintArray uiUser, jintArray uiItem,
                                                                        jintArray ctxPtr, jintArray ctxConds) {
    const std::vector<int32_t> rp = ints(env, rowPtr), ci = ints(env, colInd), uu = ints(env, uiUser), ui = ints(env, uiItem),
                               cp = ints(env, ctxPtr), cc = ints(env, ctxConds);
    const std::vector<double> d = doubles(env, data);
    if (bad_csr(env, rp, ci.size(), d.size(), &uu, &ui)) return;
    if (bad_args(env, cp.empty() || cp.front() != 0 || (std::size_t)cp.back() != cc.size(), "ctxPtr must start at 0 and end at ctxConds.length")) return;
    std::vector<int32_t> u, j;
    expand_pairs(rp, uu, ui, ci.size(), u, j);
    throw_group(env, (cmi_group_handle)g,
                cmi_group_set_ratings((cmi_group_handle)g, (int64_t)ci.size(), u.data(), j.data(), ci.data(), d.data(), (int32_t)cp.size() - 1,
                                      cp.data(), cc.data()));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupSetRatings2D(JNIEnv *env, jclass, jlong g, jintArray rowPtr, jintArray colInd,
                                                                       jdoubleArray data) {
    const std::vector<int32_t> rp = ints(env, rowPtr), ci = ints(env, colInd);
    const std::vector<double> d = doubles(env, data);
    if (bad_csr(env, rp, ci.size(), d.size(), nullptr, nullptr)) return;
    std::vector<int32_t> u(ci.size());
    for (std::size_t r = 0; r + 1 < rp.size(); ++r)
        for (int32_t q = rp[r]; q < rp[r + 1]; ++q) u[(std::size_t)q] = (int32_t)r;
    throw_group(env, (cmi_group_handle)g,
                cmi_group_set_ratings((cmi_group_handle)g, (int64_t)ci.size(), u.data(), ci.data(), nullptr, d.data(), 0, nullptr, nullptr));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupSetMatrix(JNIEnv *env, jclass, jlong g, jint which, jobjectArray rows) {
    jsize nr = 0, nc = 0;
    const std::vector<double> flat = flatten(env, rows, &nr, &nc);
    throw_group(env, (cmi_group_handle)g, cmi_group_set_state((cmi_group_handle)g, which, flat.data(), (int64_t)flat.size(), CMI_DTYPE_F64));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupGetMatrix(JNIEnv *env, jclass, jlong g, jint which, jobjectArray rows) {
    jsize nr = 0, nc = 0;
    std::vector<double> flat = flatten(env, rows, &nr, &nc); // for the shape
    const int rc = cmi_group_get_state((cmi_group_handle)g, which, flat.data(), (int64_t)flat.size(), CMI_DTYPE_F64);
    if (rc != CMI_OK) return throw_group(env, (cmi_group_handle)g, rc);
    scatter(env, rows, flat, nr, nc);
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupSetVector(JNIEnv *env, jclass, jlong g, jint which, jdoubleArray v) {
    const std::vector<double> p = doubles(env, v);
    throw_group(env, (cmi_group_handle)g, cmi_group_set_state((cmi_group_handle)g, which, p.data(), (int64_t)p.size(), CMI_DTYPE_F64));
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupGetVector(JNIEnv *env, jclass, jlong g, jint which, jdoubleArray v) {
    std::vector<double> p((std::size_t)env->GetArrayLength(v));
    const int rc = cmi_group_get_state((cmi_group_handle)g, which, p.data(), (int64_t)p.size(), CMI_DTYPE_F64);
    if (rc != CMI_OK) return throw_group(env, (cmi_group_handle)g, rc);
    env->SetDoubleArrayRegion(v, 0, (jsize)p.size(), p.data());
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupSetLrScale(JNIEnv *env, jclass, jlong g, jdouble scale) {
    throw_group(env, (cmi_group_handle)g, cmi_group_set_lr_scale((cmi_group_handle)g, scale));
}

JNIEXPORT jdouble JNICALL Java_carskit_alg_gpu_NativeMF_groupTrainEpoch(JNIEnv *env, jclass, jlong g, jdouble lRate) {
    double loss = 0;
    throw_group(env, (cmi_group_handle)g, cmi_group_train_epoch((cmi_group_handle)g, lRate, &loss));
    return loss;
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_groupEvalRatings(JNIEnv *env, jclass, jlong g, jintArray u, jintArray j,
                                                                              jintArray ctx, jdoubleArray r, jdouble minRate,
                                                                              jdouble maxRate) {
    const std::vector<int32_t> pu = ints(env, u), pj = ints(env, j), pc = ints(env, ctx);
    const std::vector<double> pr = doubles(env, r);
    if (bad_tuples(env, pu.size(), pj.size(), ctx != nullptr, pc.size(), true, pr.size())) return nullptr;
    double out[6] = {0, 0, 0, 0, 0, 0};
    int64_t cnt = 0;
    const int rc = cmi_group_eval_ratings((cmi_group_handle)g, (int64_t)pu.size(), pu.data(), pj.data(), ctx ? pc.data() : nullptr, pr.data(),
                                          minRate, maxRate, out, &cnt);
    out[5] = (double)cnt;
    if (rc != CMI_OK) {
        throw_group(env, (cmi_group_handle)g, rc);
        return nullptr;
    }
    return to_java(env, out, 6);
}

JNIEXPORT void JNICALL Java_carskit_alg_gpu_NativeMF_groupSetEvalRatings(JNIEnv *env, jclass, jlong g, jintArray u, jintArray j, jintArray ctx,
                                                                         jdoubleArray r) {
    const std::vector<int32_t> pu = ints(env, u), pj = ints(env, j), pc = ints(env, ctx);
    const std::vector<double> pr = doubles(env, r);
    if (bad_tuples(env, pu.size(), pj.size(), ctx != nullptr, pc.size(), true, pr.size())) return;
    throw_group(env, (cmi_group_handle)g,
                cmi_group_set_eval_ratings((cmi_group_handle)g, (int64_t)pu.size(), pu.data(), pj.data(), ctx ? pc.data() : nullptr, pr.data()));
}

JNIEXPORT jdoubleArray JNICALL Java_carskit_alg_gpu_NativeMF_groupEvalResident(JNIEnv *env, jclass, jlong g, jdouble minRate, jdouble maxRate) {
    double out[6] = {0, 0, 0, 0, 0, 0};
    int64_t cnt = 0;
    const int rc = cmi_group_eval_resident((cmi_group_handle)g, minRate, maxRate, out, &cnt);
    out[5] = (double)cnt;
    if (rc != CMI_OK) {
        throw_group(env, (cmi_group_handle)g, rc);
        return nullptr;
    }
    return to_java(env, out, 6);
}

} // extern "C"
