// recommender.hpp -- C++ host layer above the C ABI, mirroring the reference's plugin interface for the accelerated
// path (same class names, hooks and lifecycle as src/carskit/generic/{Recommender,IterativeRecommender,
// ContextRecommender}.java and the model classes).  It uses ONLY include/carskit_mi355x.h -- it is what a JVM-less
// deployment links, and it shows the boundary is sufficient for a complete host.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/carskit_mi355x.h"
#include "configer.hpp"

namespace carskit {

// StrictMath.log = fdlibm __ieee754_log (published algorithm), for finite positive normal x: glibc's log is correctly
// rounded and differs from fdlibm in the last ulp for some arguments, which would desynchronise nextGaussian().
inline double fdlibm_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    static_assert(sizeof bits == sizeof x, "");
    std::memcpy(&bits, &x, 8);
    int32_t hx = (int32_t)(bits >> 32), k = 0;
    if (hx < 0x00100000 || hx >= 0x7ff00000) return std::log(x);
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i0 = (hx + 0x95f64) & 0x100000;
    bits = ((uint64_t)(uint32_t)(hx | (i0 ^ 0x3ff00000)) << 32) | (bits & 0xffffffffULL);
    std::memcpy(&x, &bits, 8);
    k += i0 >> 20;
    const double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6)), t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    if (((hx - 0x6147a) | (0x6b851 - hx)) > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// java.util.Random (published algorithm): fold assignment follows the reference's stream
// (happy.coding.math.Randoms.seed(n) -> new Random(n); uniform() -> nextDouble()).
class JavaRandom {
  public:
    explicit JavaRandom(int64_t seed) : seed_(((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int32_t next(int bits) {
        seed_ = (seed_ * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)((int64_t)seed_ >> (48 - bits));
    }
    double nextDouble() { return (double)(((int64_t)next(26) << 27) + next(27)) * 0x1.0p-53; }
    double nextGaussian() { // polar method with StrictMath.log / StrictMath.sqrt
        if (have_) {
            have_ = false;
            return cached_;
        }
        double v1, v2, s;
        do {
            v1 = 2 * nextDouble() - 1;
            v2 = 2 * nextDouble() - 1;
            s = v1 * v1 + v2 * v2;
        } while (s >= 1 || s == 0);
        const double m = std::sqrt(-2 * fdlibm_log(s) / s);
        cached_ = v2 * m;
        have_ = true;
        return v1 * m;
    }

  private:
    uint64_t seed_;
    double cached_ = 0;
    bool have_ = false;
};

// The (user-item x context) rating matrix as flat tuples in MatrixIterator (CRS) order + the id-space sizes.
struct RatingData {
    int32_t n_users = 0, n_items = 0, n_conds = 0, n_dims = 0;
    std::vector<int32_t> u, j, ctx;
    std::vector<double> r;
    std::vector<int32_t> ctx_ptr, ctx_conds;
    std::vector<int32_t> empty_conds; // EmptyContextConditions (DataDAO.java:213-214): the ":na" conditions in header order
    double min_rate = 1, max_rate = 5;
    int64_t n() const { return (int64_t)r.size(); }
    RatingData subset(const std::vector<int64_t> &idx) const {
        RatingData d = *this;
        d.u.clear(), d.j.clear(), d.ctx.clear(), d.r.clear();
        for (int64_t t : idx) {
            d.u.push_back(u[(size_t)t]);
            d.j.push_back(j[(size_t)t]);
            d.ctx.push_back(ctx[(size_t)t]);
            d.r.push_back(r[(size_t)t]);
        }
        return d;
    }
};

// DataSplitter.splitFolds (src/carskit/data/processor/DataSplitter.java:102-133): fold label 1..k per matrix entry
inline std::vector<int> split_folds(int64_t n, int k_fold, int64_t seed, int *num_fold) {
    const int nf = (int64_t)k_fold > n ? (int)n : k_fold;
    JavaRandom rnd(seed);
    std::vector<double> rdm((size_t)n);
    std::vector<int> fold((size_t)n);
    const double indv = ((double)n + 0.0) / nf;
    for (int64_t i = 0; i < n; ++i) {
        rdm[(size_t)i] = rnd.nextDouble();
        fold[(size_t)i] = (int)((double)i / indv) + 1;
    }
    std::vector<int64_t> ord((size_t)n);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return rdm[(size_t)a] < rdm[(size_t)b]; });
    std::vector<int> out((size_t)n);
    for (int64_t i = 0; i < n; ++i) out[(size_t)i] = fold[(size_t)ord[(size_t)i]];
    *num_fold = nf;
    return out;
}

// static hyper-parameters (Recommender.java:194-247, IterativeRecommender.java:80-103, FM.java:53-54)
struct Conf {
    int numFactors = 10, numIters = 100;
    double initLRate = java_float("0.01"), maxLRate = -1, decay = -1;
    bool isBoldDriver = false, verbose = true;
    double regU = java_float("0.01"), regI = regU, regB = regU, regC = regU, regLw = 0, regLf = 0;
    std::string earlyStop; // "", "Loss", "MAE", "RMSE"
    int64_t randSeed = 1;
    unsigned flags = 0;
    int device = 0;
    int deviceShare = 1;       // recommenders training concurrently on `device` (`cv -p on`: the folds that landed on it): cmi_set_device_share
    int shards = 1;            // --shards N: ONE recommender sharded by user over N GPUs (cmi_group_*; the Java host's -Dcarskit.shards)
    double shardsLrScale = 0;  // local learning-rate scale of a sharded run; 0 = sqrt(shards)
    // item.ranking / ratings.setup -threshold / eval.strategy (Recommender.java:211-217,242; CARSKit.java:262)
    bool isRankingPred = false;
    int numRecs = 10, numIgnore = -1;
    double binThold = -1.0;
    std::string evalStrategy = "ucu";
    int numF = 10; // `-f` of the recommender line (CAMF_LCS.java:37: algoOptions.getInt("-f", 10))
    // `output.setup ... --save-model` (Recommender.java:240,364-365) and the reference's loadModel() branch of execute()
    // (Recommender.java:332-338: reachable only with Debug.OFF there; here the driver's --load-model flag)
    bool isSaveModel = false, loadModel = false;
    std::string workingPath;
    Conf() {}
    explicit Conf(const FileConfiger &cf) {
        if (cf.contains("learn.rate")) {
            LineConfiger lc = cf.getParamOptions("learn.rate");
            initLRate = java_float(lc.getMainParam());
            maxLRate = lc.getFloat("-max", -1);
            isBoldDriver = lc.contains("-bold-driver");
            decay = lc.getFloat("-decay", -1);
        }
        if (cf.contains("reg.lambda")) {
            LineConfiger ro = cf.getParamOptions("reg.lambda");
            const double reg = java_float(ro.getMainParam());
            regU = ro.getFloat("-u", reg);
            regI = ro.getFloat("-i", reg);
            regB = ro.getFloat("-b", reg);
            regC = ro.getFloat("-c", reg);
        }
        numFactors = cf.getInt("num.factors", 10);
        numIters = cf.getInt("num.max.iter", 100);
        if (cf.contains("evaluation.setup")) {
            LineConfiger ev = cf.getParamOptions("evaluation.setup");
            const std::string es = lower(ev.getString("--early-stop"));
            earlyStop = es == "loss" ? "Loss" : es == "mae" ? "MAE" : es == "rmse" ? "RMSE" : "";
            randSeed = ev.getLong("--rand-seed", 1);
        }
        if (cf.contains("output.setup")) {
            verbose = cf.getParamOptions("output.setup").isOn("-verbose", true);
            isSaveModel = cf.getParamOptions("output.setup").contains("--save-model");
        }
        if (cf.contains("item.ranking")) {
            LineConfiger rk = cf.getParamOptions("item.ranking");
            isRankingPred = rk.isMainOn();
            numRecs = rk.getInt("-topN", -1);
            if (numRecs < 0) numRecs = 10;
            numIgnore = rk.getInt("-ignore", -1);
            if (rk.contains("-diverse")) throw std::runtime_error("item.ranking -diverse is not on the accelerated path");
        }
        if (cf.contains("ratings.setup")) binThold = cf.getParamOptions("ratings.setup").getFloat("-threshold", -1);
        evalStrategy = lower(cf.getString("eval.strategy", "ucu"));
        if (cf.contains("recommender")) numF = cf.getParamOptions("recommender").getInt("-f", 10);
        if (cf.contains("FM")) {
            LineConfiger fm = cf.getParamOptions("FM");
            regLw = fm.getFloat("-lw", 0);
            regLf = fm.getFloat("-lf", 0);
        }
    }
};

typedef std::map<std::string, double> Measures;
typedef std::function<void(const std::string &)> Logger;

inline void check(int rc, cmi_handle h, const char *what) {
    if (rc != CMI_OK) throw std::runtime_error(std::string(what) + ": " + cmi_last_error(h));
}

// carskit.generic.Recommender + IterativeRecommender (+ ContextRecommender), for the models libcarskit_mi355x runs
class IterativeRecommender {
  public:
    IterativeRecommender(int model, const char *name, bool is_cars, const RatingData &train, const RatingData &test, int fold,
                         const Conf &conf, Logger log)
        : algoName(name), model_(model), isCARS_(is_cars), trainMatrix(train), testMatrix(test), fold_(fold), conf_(conf),
          log_(log) {
        lRate = conf.initLRate;
        double s = 0;
        int64_t cnt = 0;
        for (double v : train.r) { // SparseMatrix.getGlobalAvg: sum / #non-zero
            s += v;
            if (v != 0.0) ++cnt;
        }
        globalMean = cnt ? s / (double)cnt : std::nan("");
    }
    virtual ~IterativeRecommender() {
        if (g_) cmi_group_destroy(g_);
        if (h_) cmi_destroy(h_);
    }

    // ---- hooks, named as in the reference ---------------------------------------------------------------------
    virtual void initModel() { // IterativeRecommender.java:232-247 + the model class: P, Q ~ N(0,0.1), then the biases
        JavaRandom rnd(conf_.randSeed);   // the reference's init stream is unseeded: any stream is as faithful
        auto gauss = [&](std::vector<double> &v, size_t n) {
            v.resize(n);
            for (double &x : v) x = 0.0 + 0.1 * rnd.nextGaussian();
        };
        auto unif = [&](std::vector<double> &v, size_t n) {
            v.resize(n);
            for (double &x : v) x = rnd.nextDouble();
        };
        const size_t nu = (size_t)trainMatrix.n_users, ni = (size_t)trainMatrix.n_items, nc = (size_t)trainMatrix.n_conds,
                     k = (size_t)conf_.numFactors;
        gauss(state[CMI_STATE_P], nu * k);
        gauss(state[CMI_STATE_Q], ni * k);
        switch (model_) {
        case CMI_MODEL_BIASEDMF:
            gauss(state[CMI_STATE_USER_BIAS], nu);
            gauss(state[CMI_STATE_ITEM_BIAS], ni);
            break;
        case CMI_MODEL_CAMF_C:
            gauss(state[CMI_STATE_USER_BIAS], nu);
            gauss(state[CMI_STATE_ITEM_BIAS], ni);
            gauss(state[CMI_STATE_COND_BIAS], nc);
            break;
        case CMI_MODEL_CAMF_CI:
            gauss(state[CMI_STATE_USER_BIAS], nu);
            unif(state[CMI_STATE_IC_BIAS], ni * nc); // icBias.init() = uniform(0,1), CAMF_CI.java:60
            break;
        case CMI_MODEL_CAMF_CU:
            gauss(state[CMI_STATE_ITEM_BIAS], ni);
            unif(state[CMI_STATE_UC_BIAS], nu * nc);
            break;
        case CMI_MODEL_CAMF_CUCI:
            gauss(state[CMI_STATE_UC_BIAS], nu * nc);
            gauss(state[CMI_STATE_IC_BIAS], ni * nc);
            break;
        case CMI_MODEL_SVDPP: // SVDPlusPlus.java:46-53: BiasedMF.initModel, then Y.init(initMean, initStd)
            gauss(state[CMI_STATE_USER_BIAS], nu);
            gauss(state[CMI_STATE_ITEM_BIAS], ni);
            gauss(state[CMI_STATE_Y], ni * k);
            break;
        case CMI_MODEL_CAMF_ICS: // CAMF_ICS.java:36-51: isRankingPred -> P.init(), Q.init() (uniform) on top of the gaussian draws
            unif(state[CMI_STATE_P], nu * k);
            unif(state[CMI_STATE_Q], ni * k);
            state[CMI_STATE_CC_MATRIX].assign(nc * nc, 1.0);
            break;
        case CMI_MODEL_CAMF_LCS: // CAMF_LCS.java:34-41: cfMatrix_LCS.init() = uniform(0,1)
            unif(state[CMI_STATE_CF_MATRIX], nc * (size_t)conf_.numF);
            break;
        case CMI_MODEL_CAMF_MCS: { // CAMF_MCS.java:41-52: cVector_MCS.init(upbound) = uniform(0, 1/sqrt(numContextDims))
            const double up = 1.0 / std::sqrt((double)std::max(1, trainMatrix.n_dims));
            unif(state[CMI_STATE_C_VECTOR], nc);
            for (double &x : state[CMI_STATE_C_VECTOR]) x *= up;
            break;
        }
        default: break;
        }
    }

    // --shards N (N > 1, models whose ratings commute): the epochs run on a cmi_group -- the library cuts the ratings by user, runs the
    // shards' epochs concurrently and merges the item-side moves (RCCL reduce-scatter + all-gather, or in-process when shards share a
    // device) -- steered by the unchanged isConverged(); `--early-stop MAE|RMSE` scores the shards' resident test tuples.  Afterwards the
    // model is copied back and lives on in a plain handle, so evaluation / ranking / --save-model are the single-GPU code below.
    void trainSharded(unsigned flags) {
        auto gcheck = [&](int rc, const char *what) {
            if (rc != CMI_OK) throw std::runtime_error(std::string(what) + ": " + cmi_group_last_error(g_));
        };
        int rc = cmi_group_create(model_, conf_.numFactors, trainMatrix.n_users, trainMatrix.n_items, trainMatrix.n_conds, conf_.shards, nullptr,
                                  flags, &g_);
        if (rc != CMI_OK) throw std::runtime_error(std::string("cmi_group_create: ") + cmi_group_last_error(nullptr));
        gcheck(cmi_group_set_hparams(g_, conf_.regU, conf_.regI, conf_.regB, conf_.regC, globalMean), "cmi_group_set_hparams");
        if (isCARS_) {
            gcheck(cmi_group_set_ratings(g_, trainMatrix.n(), trainMatrix.u.data(), trainMatrix.j.data(), trainMatrix.ctx.data(),
                                         trainMatrix.r.data(), (int32_t)trainMatrix.ctx_ptr.size() - 1, trainMatrix.ctx_ptr.data(),
                                         trainMatrix.ctx_conds.data()),
                   "cmi_group_set_ratings");
        } else {
            std::vector<int32_t> u2, j2;
            std::vector<double> r2;
            to2d(trainMatrix, u2, j2, r2);
            gcheck(cmi_group_set_ratings(g_, (int64_t)r2.size(), u2.data(), j2.data(), nullptr, r2.data(), 0, nullptr, nullptr), "cmi_group_set_ratings");
        }
        for (auto &kv : state) gcheck(cmi_group_set_state(g_, kv.first, kv.second.data(), (int64_t)kv.second.size(), CMI_DTYPE_F64), "cmi_group_set_state");
        gcheck(cmi_group_set_lr_scale(g_, conf_.shardsLrScale > 0 ? conf_.shardsLrScale : std::sqrt((double)conf_.shards)), "cmi_group_set_lr_scale");
        if (conf_.earlyStop == "MAE" || conf_.earlyStop == "RMSE") {
            gcheck(cmi_group_set_eval_ratings(g_, testMatrix.n(), testMatrix.u.data(), testMatrix.j.data(), isCARS_ ? testMatrix.ctx.data() : nullptr,
                                              testMatrix.r.data()),
                   "cmi_group_set_eval_ratings");
            evalResident_ = testMatrix.n() > 0;
        }
        for (int iter = 1; iter <= conf_.numIters; ++iter) {
            gcheck(cmi_group_train_epoch(g_, lRate, &loss), "cmi_group_train_epoch");
            losses.push_back(loss);
            itersDone = iter;
            if (isConverged(iter)) break;
        }
        for (auto &kv : state) gcheck(cmi_group_get_state(g_, kv.first, kv.second.data(), (int64_t)kv.second.size(), CMI_DTYPE_F64), "cmi_group_get_state");
        cmi_group_destroy(g_);
        g_ = nullptr;
        evalResident_ = false;
    }

    virtual void buildModel() {
        const bool chained = model_ == CMI_MODEL_CAMF_C || (model_ >= CMI_MODEL_SVDPP && model_ <= CMI_MODEL_CAMF_MCS);
        unsigned flags = conf_.flags | (chained ? CMI_FLAG_SCHED_SERIAL : 0u); // one dependent chain in CRS order
        const bool sharded = conf_.shards > 1 && !chained && !conf_.loadModel;
        if (sharded) trainSharded(flags);
        int rc = cmi_create(model_, conf_.numFactors, trainMatrix.n_users, trainMatrix.n_items, trainMatrix.n_conds,
                            conf_.device, flags, &h_);
        if (rc != CMI_OK) throw std::runtime_error(std::string("cmi_create: ") + cmi_last_error(nullptr));
        check(cmi_set_hparams(h_, conf_.regU, conf_.regI, conf_.regB, conf_.regC, globalMean), h_, "cmi_set_hparams");
        if (conf_.deviceShare > 1) check(cmi_set_device_share(h_, conf_.deviceShare), h_, "cmi_set_device_share");
        if (model_ >= CMI_MODEL_CAMF_ICS && model_ <= CMI_MODEL_CAMF_MCS)
            check(cmi_set_sim_params(h_, conf_.numF, std::max(1, trainMatrix.n_dims), trainMatrix.empty_conds.data(),
                                     (int)trainMatrix.empty_conds.size()),
                  h_, "cmi_set_sim_params");
        if (isCARS_) {
            check(cmi_set_ratings(h_, trainMatrix.n(), trainMatrix.u.data(), trainMatrix.j.data(), trainMatrix.ctx.data(),
                                  trainMatrix.r.data(), (int32_t)trainMatrix.ctx_ptr.size() - 1, trainMatrix.ctx_ptr.data(),
                                  trainMatrix.ctx_conds.data()),
                  h_, "cmi_set_ratings");
        } else { // Recommender.initModel (:1076-1081): the 2-D train matrix, mean over contexts per (user,item)
            std::vector<int32_t> u2, j2;
            std::vector<double> r2;
            to2d(trainMatrix, u2, j2, r2);
            check(cmi_set_ratings(h_, (int64_t)r2.size(), u2.data(), j2.data(), nullptr, r2.data(), 0, nullptr, nullptr), h_,
                  "cmi_set_ratings");
        }
        if (*cmi_schedule_note(h_) && log_) log_(std::string("note: ") + cmi_schedule_note(h_));
        for (auto &kv : state) // copy-in
            check(cmi_set_state(h_, kv.first, kv.second.data(), (int64_t)kv.second.size(), CMI_DTYPE_F64), h_, "cmi_set_state");
        if (!sharded && (conf_.earlyStop == "MAE" || conf_.earlyStop == "RMSE")) { // evaluated after every epoch: keep the test tuples on the device
            check(cmi_set_eval_ratings(h_, testMatrix.n(), testMatrix.u.data(), testMatrix.j.data(),
                                       isCARS_ ? testMatrix.ctx.data() : nullptr, testMatrix.r.data()),
                  h_, "cmi_set_eval_ratings");
            evalResident_ = testMatrix.n() > 0;
        }
        if (conf_.loadModel) { // Recommender.execute's loadModel() branch (:332-338): the stored model instead of initModel + the epochs
            int done = 0;
            check(cmi_load_model(h_, modelPath().c_str(), &lRate, &last_loss, &done), h_, "cmi_load_model");
            itersDone = done;
            if (log_) log_("A recommender model is loaded from " + modelPath());
        } else if (!sharded) {
            for (int iter = 1; iter <= conf_.numIters; ++iter) {
                check(cmi_train_epoch(h_, lRate, &loss), h_, "cmi_train_epoch"); // the for(MatrixEntry me : trainMatrix) body
                losses.push_back(loss);
                itersDone = iter;
                if (isConverged(iter)) break;
            }
        }
        for (auto &kv : state) // copy-back
            check(cmi_get_state(h_, kv.first, kv.second.data(), (int64_t)kv.second.size(), CMI_DTYPE_F64), h_, "cmi_get_state");
    }

    virtual Measures evalRatings() { // Recommender.java:504-594 (numeric part)
        double out[5] = {0, 0, 0, 0, 0};
        int64_t cnt = 0;
        if (evalResident_ && g_) {
            if (cmi_group_eval_resident(g_, trainMatrix.min_rate, trainMatrix.max_rate, out, &cnt) != CMI_OK)
                throw std::runtime_error(std::string("cmi_group_eval_resident: ") + cmi_group_last_error(g_));
        } else if (evalResident_)
            check(cmi_eval_resident(h_, trainMatrix.min_rate, trainMatrix.max_rate, out, &cnt), h_, "cmi_eval_resident");
        else
        check(cmi_eval_ratings(h_, testMatrix.n(), testMatrix.u.data(), testMatrix.j.data(),
                               isCARS_ ? testMatrix.ctx.data() : nullptr, testMatrix.r.data(), trainMatrix.min_rate,
                               trainMatrix.max_rate, out, &cnt),
              h_, "cmi_eval_ratings");
        return Measures{{"MAE", out[0]}, {"RMSE", out[1]}, {"NMAE", out[2]}, {"rMAE", out[3]}, {"rRMSE", out[4]}, {"MPE", 0.0}};
    }

    virtual Measures evalRankings() { // Recommender.java:668-964
        static const char *names[CMI_RANK_MEASURES] = {"Pre5", "Pre10", "PreN", "Rec5", "Rec10", "RecN", "AUC5", "AUC10", "AUCN",
                                                       "MAP5", "MAP10", "MAPN", "NDCG5", "NDCG10", "NDCGN", "MRR5", "MRR10",
                                                       "MRRN", "D5", "D10", "DN"};
        double out[CMI_RANK_MEASURES];
        int64_t nq = 0;
        check(cmi_eval_rankings(h_, trainMatrix.n(), trainMatrix.u.data(), trainMatrix.j.data(), trainMatrix.ctx.data(),
                                trainMatrix.r.data(), testMatrix.n(), testMatrix.u.data(), testMatrix.j.data(),
                                testMatrix.ctx.data(), testMatrix.r.data(), conf_.binThold, conf_.numRecs, conf_.numIgnore,
                                conf_.evalStrategy == "uc" ? CMI_RANK_UC : CMI_RANK_UCU, out, &nq, nullptr, nullptr, nullptr,
                                nullptr, nullptr),
              h_, "cmi_eval_rankings");
        Measures m;
        for (int i = 0; i < CMI_RANK_MEASURES; ++i) m[names[i]] = out[i];
        return m;
    }

    Measures execute() { // Recommender.java:319-366
        auto t0 = std::chrono::steady_clock::now();
        initModel();
        buildModel();
        auto t1 = std::chrono::steady_clock::now();
        measures = conf_.isRankingPred ? evalRankings() : evalRatings(); // :346
        auto t2 = std::chrono::steady_clock::now();
        measures["TrainTime"] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        measures["TestTime"] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        if (conf_.isSaveModel) saveModel(); // :364-365
        return measures;
    }

    // <workingPath>/<algoName>/model<foldInfo>.cmi -- the reference writes <workingPath>/<algoName>/{userFactors,itemFactors,userBiases,
    // itemBiases}<foldInfo>.bin as Java object streams and forgets the context tables (IterativeRecommender.java:249-270); here ONE
    // cmi_save_model file holds every container plus the loop's resume state (lRate, last loss, epochs done)
    std::string modelPath() const {
        return conf_.workingPath + algoName + "/model" + (fold_ > 0 ? " fold [" + std::to_string(fold_) + "]" : "") + ".cmi";
    }
    void saveModel() {
        if (!h_) { // FM keeps its model behind a cmi_fm_handle, which has no persistence entry point
            if (log_) log_("--save-model: not available for " + algoName);
            return;
        }
        const std::string mk = "mkdir -p '" + conf_.workingPath + algoName + "'";
        if (std::system(mk.c_str()) != 0) throw std::runtime_error("cannot create " + conf_.workingPath + algoName);
        check(cmi_save_model(h_, modelPath().c_str(), lRate, last_loss, itersDone), h_, "cmi_save_model");
        if (log_) log_("Learned models are saved to folder \"" + conf_.workingPath + algoName + "/\"");
    }
    int itersDone = 0;

    bool isConverged(int iter) { // IterativeRecommender.java:145-199
        const float delta_loss = (float)(last_loss - loss);
        if (conf_.earlyStop == "Loss") {
            measure = loss;
            last_measure = last_loss;
        } else if (conf_.earlyStop == "MAE" || conf_.earlyStop == "RMSE") {
            measure = evalRatings()[conf_.earlyStop];
        }
        const float delta_measure = (float)(last_measure - measure);
        if (conf_.verbose && log_) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s%s iter %d: loss = %g, delta_loss = %g, learn_rate = %g", algoName.c_str(),
                     fold_ > 0 ? (" fold [" + std::to_string(fold_) + "]").c_str() : "", iter, (double)(float)loss,
                     (double)delta_loss, (double)(float)lRate);
            log_(buf);
        }
        if (std::isnan(loss) || std::isinf(loss))
            throw std::runtime_error("Loss = NaN or Infinity: current settings does not fit the recommender! Change the settings and try again!");
        const bool converged = std::fabs(loss) < 1e-5 || (delta_measure > 0 && delta_measure < 1e-5);
        if (!converged) updateLRate(iter);
        last_loss = loss;
        last_measure = measure;
        return converged;
    }

    void updateLRate(int iter) { // IterativeRecommender.java:216-229
        if (lRate <= 0) return;
        if (conf_.isBoldDriver && iter > 1) lRate = std::fabs(last_loss) > std::fabs(loss) ? lRate * 1.05 : lRate * 0.5;
        else if (conf_.decay > 0 && conf_.decay < 1) lRate *= conf_.decay;
        if (conf_.maxLRate > 0 && lRate > conf_.maxLRate) lRate = conf_.maxLRate;
    }

    static void to2d(const RatingData &d, std::vector<int32_t> &u2, std::vector<int32_t> &j2, std::vector<double> &r2) {
        // DataDAO.toTraditionalSparseMatrix: the mean of a (user, item) pair's ratings over its contexts, pairs in (user, item) order.
        // (key, tuple) pairs sorted: a cell's ratings are then added in tuple order, as a map keyed by the pair would add them.
        std::vector<std::pair<uint64_t, int64_t>> ord((size_t)d.n());
        for (int64_t t = 0; t < d.n(); ++t)
            ord[(size_t)t] = {((uint64_t)(uint32_t)d.u[(size_t)t] << 32) | (uint32_t)d.j[(size_t)t], t};
        std::sort(ord.begin(), ord.end());
        for (size_t a = 0; a < ord.size();) {
            double sum = 0.0, cnt = 0.0;
            size_t b = a;
            for (; b < ord.size() && ord[b].first == ord[a].first; ++b) {
                sum += d.r[(size_t)ord[b].second];
                cnt += 1.0;
            }
            u2.push_back((int32_t)(ord[a].first >> 32));
            j2.push_back((int32_t)(ord[a].first & 0xffffffffu));
            r2.push_back(sum / cnt);
            a = b;
        }
    }

    std::string algoName;
    Measures measures;
    bool evalResident_ = false;
    std::map<int, std::vector<double>> state; // CMI_STATE_* -> container (P, Q, userBias, ...)
    std::vector<double> losses;
    double lRate = 0, loss = 0, last_loss = 0, measure = 0, last_measure = 0, globalMean = 0;

  protected:
    int model_;
    bool isCARS_;
    RatingData trainMatrix, testMatrix;
    int fold_;
    Conf conf_;
    Logger log_;
    cmi_handle h_ = nullptr;
    cmi_group_handle g_ = nullptr; // live only inside trainSharded()
};

#define CARSKIT_MODEL(cls, id, cars) CARSKIT_NAMED_MODEL(cls, #cls, id, cars)
#define CARSKIT_NAMED_MODEL(cls, name, id, cars)                                                                      \
    class cls : public IterativeRecommender {                                                                          \
      public:                                                                                                          \
        cls(const RatingData &tr, const RatingData &te, int fold, const Conf &c, Logger log = nullptr)                 \
            : IterativeRecommender(id, name, cars, tr, te, fold, c, log) {}                                            \
    };
CARSKIT_MODEL(BiasedMF, CMI_MODEL_BIASEDMF, false)  // src/carskit/alg/baseline/cf/BiasedMF.java
CARSKIT_MODEL(PMF, CMI_MODEL_PMF, false)            // src/carskit/alg/baseline/cf/PMF.java
CARSKIT_MODEL(CAMF_C, CMI_MODEL_CAMF_C, true)       // src/carskit/alg/cars/adaptation/dependent/dev/CAMF_C.java
CARSKIT_MODEL(CAMF_CI, CMI_MODEL_CAMF_CI, true)     // .../dev/CAMF_CI.java
CARSKIT_MODEL(CAMF_CU, CMI_MODEL_CAMF_CU, true)     // .../dev/CAMF_CU.java
CARSKIT_MODEL(CAMF_CUCI, CMI_MODEL_CAMF_CUCI, true) // .../dev/CAMF_CUCI.java
CARSKIT_NAMED_MODEL(SVDPlusPlus, "SVD++", CMI_MODEL_SVDPP, false) // src/carskit/alg/baseline/cf/SVDPlusPlus.java:42 (algoName = "SVD++"; 2-D train matrix)
#undef CARSKIT_MODEL
#undef CARSKIT_NAMED_MODEL

// The similarity-based CAMF models are top-N recommenders: their constructors set the static isRankingPred = true
// (CAMF_ICS.java:31, CAMF_LCS.java:31, CAMF_MCS.java:37), so execute() evaluates with evalRankings() whatever item.ranking says.
#define CARSKIT_SIM_MODEL(cls, id)                                                                                     \
    class cls : public IterativeRecommender {                                                                          \
      public:                                                                                                          \
        cls(const RatingData &tr, const RatingData &te, int fold, const Conf &c, Logger log = nullptr)                 \
            : IterativeRecommender(id, #cls, true, tr, te, fold, c, log) {                                             \
            conf_.isRankingPred = true;                                                                                \
        }                                                                                                              \
    };
CARSKIT_SIM_MODEL(CAMF_ICS, CMI_MODEL_CAMF_ICS) // src/carskit/alg/cars/adaptation/dependent/sim/CAMF_ICS.java
CARSKIT_SIM_MODEL(CAMF_LCS, CMI_MODEL_CAMF_LCS) // .../sim/CAMF_LCS.java
CARSKIT_SIM_MODEL(CAMF_MCS, CMI_MODEL_CAMF_MCS) // .../sim/CAMF_MCS.java
#undef CARSKIT_SIM_MODEL

// src/carskit/alg/cars/adaptation/dependent/FM.java: w0 = 0, w ~ U(0,1), V ~ N(0,0.1) (:65-70); numIters ALS sweeps, no
// convergence check; evaluation through the generic evalRatings recipe on bounded predictions.
class FM : public IterativeRecommender {
  public:
    FM(const RatingData &tr, const RatingData &te, int fold, const Conf &c, Logger log = nullptr)
        : IterativeRecommender(-1, "FM", true, tr, te, fold, c, log) {}
    ~FM() override {
        if (fm_) cmi_fm_destroy(fm_);
    }
    void initModel() override {
        JavaRandom rnd(conf_.randSeed);
        const size_t p = (size_t)trainMatrix.n_users + trainMatrix.n_items + trainMatrix.n_conds;
        w.resize(p);
        for (double &x : w) x = rnd.nextDouble();
        V.resize(p * (size_t)conf_.numFactors);
        for (double &x : V) x = 0.0 + 0.1 * rnd.nextGaussian();
        w0 = 0.0;
    }
    void buildModel() override {
        if (cmi_fm_create(conf_.numFactors, trainMatrix.n_users, trainMatrix.n_items, trainMatrix.n_conds,
                          std::max(1, trainMatrix.n_dims), conf_.device, 0, &fm_) != CMI_OK)
            throw std::runtime_error(std::string("cmi_fm_create: ") + cmi_fm_last_error(nullptr));
        fmcheck(cmi_fm_set_hparams(fm_, conf_.regLw, conf_.regLf, 0), "cmi_fm_set_hparams");
        fmcheck(cmi_fm_set_ratings(fm_, trainMatrix.n(), trainMatrix.u.data(), trainMatrix.j.data(), trainMatrix.ctx.data(),
                                   trainMatrix.r.data()), "cmi_fm_set_ratings");
        fmcheck(cmi_fm_set_model(fm_, w0, w.data(), V.data()), "cmi_fm_set_model");
        fmcheck(cmi_fm_train(fm_, conf_.numIters), "cmi_fm_train");
        fmcheck(cmi_fm_get_model(fm_, &w0, w.data(), V.data()), "cmi_fm_get_model");
    }
    Measures evalRankings() override { // Recommender.java:668-964 with FM.predict
        static const char *names[CMI_RANK_MEASURES] = {"Pre5", "Pre10", "PreN", "Rec5", "Rec10", "RecN", "AUC5", "AUC10", "AUCN",
                                                       "MAP5", "MAP10", "MAPN", "NDCG5", "NDCG10", "NDCGN", "MRR5", "MRR10",
                                                       "MRRN", "D5", "D10", "DN"};
        double out[CMI_RANK_MEASURES];
        int64_t nq = 0;
        if (cmi_fm_eval_rankings(fm_, trainMatrix.n(), trainMatrix.u.data(), trainMatrix.j.data(), trainMatrix.ctx.data(),
                                 trainMatrix.r.data(), testMatrix.n(), testMatrix.u.data(), testMatrix.j.data(),
                                 testMatrix.ctx.data(), testMatrix.r.data(), conf_.binThold, conf_.numRecs, conf_.numIgnore,
                                 conf_.evalStrategy == "uc" ? CMI_RANK_UC : CMI_RANK_UCU, out, &nq, nullptr, nullptr, nullptr,
                                 nullptr, nullptr) != CMI_OK)
            throw std::runtime_error(std::string("cmi_fm_eval_rankings: ") + cmi_fm_last_error(fm_));
        Measures m;
        for (int i = 0; i < CMI_RANK_MEASURES; ++i) m[names[i]] = out[i];
        return m;
    }
    Measures evalRatings() override {
        std::vector<double> pred((size_t)testMatrix.n());
        fmcheck(cmi_fm_predict_batch(fm_, testMatrix.n(), testMatrix.u.data(), testMatrix.j.data(), testMatrix.ctx.data(), 1,
                                     trainMatrix.min_rate, trainMatrix.max_rate, pred.data()), "cmi_fm_predict_batch");
        double sa = 0, ss = 0, sra = 0, srs = 0;
        int64_t n = 0;
        for (size_t t = 0; t < pred.size(); ++t) {
            if (std::isnan(pred[t])) continue;
            const double rp = std::floor(pred[t] / trainMatrix.min_rate + 0.5) * trainMatrix.min_rate;
            const double e = std::fabs(testMatrix.r[t] - pred[t]), re = std::fabs(testMatrix.r[t] - rp);
            sa += e, ss += e * e, sra += re, srs += re * re;
            ++n;
        }
        const double mae = sa / (double)n;
        return Measures{{"MAE", mae}, {"RMSE", std::sqrt(ss / (double)n)}, {"NMAE", mae / (trainMatrix.max_rate - trainMatrix.min_rate)},
                        {"rMAE", sra / (double)n}, {"rRMSE", std::sqrt(srs / (double)n)}, {"MPE", 0.0}};
    }
    double w0 = 0;
    std::vector<double> w, V;

  private:
    void fmcheck(int rc, const char *what) {
        if (rc != CMI_OK) throw std::runtime_error(std::string(what) + ": " + cmi_fm_last_error(fm_));
    }
    cmi_fm_handle fm_ = nullptr;
};

// the factory switch of CARSKit.getRecommender (src/carskit/main/CARSKit.java:461-469,700-712,742), lower-cased names
inline std::unique_ptr<IterativeRecommender> getRecommender(const std::string &name, const RatingData &tr, const RatingData &te,
                                                            int fold, const Conf &c, Logger log) {
    const std::string n = lower(name);
    if (n == "biasedmf") return std::unique_ptr<IterativeRecommender>(new BiasedMF(tr, te, fold, c, log));
    if (n == "pmf") return std::unique_ptr<IterativeRecommender>(new PMF(tr, te, fold, c, log));
    if (n == "camf_c") return std::unique_ptr<IterativeRecommender>(new CAMF_C(tr, te, fold, c, log));
    if (n == "camf_ci") return std::unique_ptr<IterativeRecommender>(new CAMF_CI(tr, te, fold, c, log));
    if (n == "camf_cu") return std::unique_ptr<IterativeRecommender>(new CAMF_CU(tr, te, fold, c, log));
    if (n == "camf_cuci") return std::unique_ptr<IterativeRecommender>(new CAMF_CUCI(tr, te, fold, c, log));
    if (n == "fm") return std::unique_ptr<IterativeRecommender>(new FM(tr, te, fold, c, log));
    if (n == "svd++") return std::unique_ptr<IterativeRecommender>(new SVDPlusPlus(tr, te, fold, c, log));   // CARSKit.java:469
    if (n == "camf_ics") return std::unique_ptr<IterativeRecommender>(new CAMF_ICS(tr, te, fold, c, log));   // CARSKit.java:708
    if (n == "camf_lcs") return std::unique_ptr<IterativeRecommender>(new CAMF_LCS(tr, te, fold, c, log));   // CARSKit.java:710
    if (n == "camf_mcs") return std::unique_ptr<IterativeRecommender>(new CAMF_MCS(tr, te, fold, c, log));   // CARSKit.java:712
    throw std::runtime_error("recommender '" + name + "' is not on the accelerated path (biasedmf, pmf, svd++, camf_c, camf_ci, camf_cu, camf_cuci, camf_ics, camf_lcs, camf_mcs, fm)");
}

} // namespace carskit
