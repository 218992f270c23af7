// carskit_main.cpp -- `carskit-mi355x -c setting.conf`: the reference driver's flow (src/carskit/main/CARSKit.java:
// execute :109, preset :140, readData :220, runAlgorithm :310, runCrossValidation :388, printEvalInfo :362) for the
// recommenders libcarskit_mi355x accelerates (rating prediction, or top-N evaluation with item.ranking=on).  Links nothing but the C ABI.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>

#include "recommender.hpp"

using namespace carskit;

static std::string dirname_of(const std::string &p) {
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string("./") : p.substr(0, s + 1);
}

struct Dao {
    cmi_dao_handle h = nullptr;
    int64_t counts[8] = {};
    explicit Dao(const std::string &path, cmi_dao_handle train = nullptr) {
        const int rc = train ? cmi_dao_read_shared(path.c_str(), train, &h) : cmi_dao_read(path.c_str(), &h);
        if (rc != CMI_OK) throw std::runtime_error("DataDAO: " + std::string(cmi_dao_last_error(nullptr)));
        cmi_dao_counts(h, counts);
    }
    ~Dao() { cmi_dao_destroy(h); }
    RatingData ratingData() const {
        RatingData d;
        d.n_users = (int32_t)counts[0];
        d.n_items = (int32_t)counts[1];
        d.n_conds = (int32_t)counts[4];
        d.n_dims = (int32_t)counts[5];
        const size_t n = (size_t)counts[7], nui = (size_t)counts[2], nctx = (size_t)counts[3];
        std::vector<int32_t> ui(n), uiu(nui), uii(nui);
        d.ctx.resize(n);
        d.r.resize(n);
        cmi_dao_matrix(h, ui.data(), d.ctx.data(), d.r.data());
        cmi_dao_ui_maps(h, uiu.data(), uii.data());
        d.u.resize(n);
        d.j.resize(n);
        for (size_t t = 0; t < n; ++t) {
            d.u[t] = uiu[(size_t)ui[t]];
            d.j[t] = uii[(size_t)ui[t]];
        }
        d.ctx_ptr.resize(nctx + 1);
        d.ctx_conds.resize((size_t)cmi_dao_ctx_nnz(h));
        cmi_dao_ctx_table(h, d.ctx_ptr.data(), d.ctx_conds.data());
        {
            std::vector<int32_t> cond_dim((size_t)std::max<int64_t>(1, counts[4])), empty((size_t)std::max<int64_t>(1, counts[4]));
            int32_t n_empty = 0;
            cmi_dao_cond_info(h, cond_dim.data(), empty.data(), &n_empty);
            d.empty_conds.assign(empty.begin(), empty.begin() + n_empty);
        }
        int32_t ns = 0;
        cmi_dao_rating_scale(h, nullptr, 0, &ns);
        std::vector<double> scale((size_t)ns);
        cmi_dao_rating_scale(h, scale.data(), ns, &ns);
        if (ns > 0) {
            d.min_rate = scale.front();
            d.max_rate = scale.back();
        }
        return d;
    }
};

static std::string evalInfo(const Measures &m, const Conf &conf) { // Recommender.getEvalInfo (Recommender.java:437-499)
    char buf[1024];
    if (conf.isRankingPred) { // the reference's separators are irregular; kept as they are
        const std::string n = std::to_string(conf.numRecs);
        if (conf.numRecs != 10) {
            const std::string fmt = "Pre5: %.6f,Pre10: %.6f, Pre" + n + ": %.6f, Rec5: %.6f, Rec10: %.6f, Rec" + n + ": %.6f, " +
                                    "AUC5: %.6f, AUC10: %.6f, AUC" + n + ": %.6f, MAP5: %.6f, MAP10: %.6f, MAP" + n + ": %.6f, " +
                                    "NDCG5: %.6f, NDCG10: %.6f,NDCG" + n + ": %.6f,MRR5: %.6f, MRR10: %.6f,MRR" + n + ": %.6f";
            snprintf(buf, sizeof buf, fmt.c_str(), m.at("Pre5"), m.at("Pre10"), m.at("PreN"), m.at("Rec5"), m.at("Rec10"), m.at("RecN"),
                     m.at("AUC5"), m.at("AUC10"), m.at("AUCN"), m.at("MAP5"), m.at("MAP10"), m.at("MAPN"), m.at("NDCG5"),
                     m.at("NDCG10"), m.at("NDCGN"), m.at("MRR5"), m.at("MRR10"), m.at("MRRN"));
        } else {
            snprintf(buf, sizeof buf,
                     "Pre5: %.6f,Pre10: %.6f, Rec5: %.6f, Rec10: %.6f, AUC5: %.6f, AUC10: %.6f, MAP5: %.6f, MAP10: %.6f,"
                     "NDCG5: %.6f, NDCG10: %.6f,MRR5: %.6f, MRR10: %.6f",
                     m.at("Pre5"), m.at("Pre10"), m.at("Rec5"), m.at("Rec10"), m.at("AUC5"), m.at("AUC10"), m.at("MAP5"),
                     m.at("MAP10"), m.at("NDCG5"), m.at("NDCG10"), m.at("MRR5"), m.at("MRR10"));
        }
        return buf;
    }
    snprintf(buf, sizeof buf, "MAE: %.6f, RMSE: %.6f, NAME: %.6f, rMAE: %.6f, rRMSE: %.6f, MPE: %.6f", m.at("MAE"), m.at("RMSE"),
             m.at("NMAE"), m.at("rMAE"), m.at("rRMSE"), m.at("MPE"));
    return buf;
}

static int run(const std::string &config, unsigned flags, int iters_override, bool precise, bool load_model, int shards, double shards_lrscale) {
    Logger log = [](const std::string &s) { std::cout << s << std::endl; };
    FileConfiger cf(config);
    Conf conf(cf);
    conf.flags = flags;
    if (iters_override > 0) conf.numIters = iters_override;
    // preset + readData
    const std::string ratingFile = cf.getPath("dataset.ratings");
    if (ratingFile.empty() || !std::ifstream(ratingFile)) throw std::runtime_error("Your rating file path is incorrect: File doesn't exist. Please double check your configuration.");
    LineConfiger out = cf.getParamOptions("output.setup");
    const std::string work = dirname_of(ratingFile) + out.getString("-folder", "CARSKit.Workspace") + "/";
    std::string mk = "mkdir -p '" + work + "'";
    if (std::system(mk.c_str()) != 0) throw std::runtime_error("cannot create " + work);
    log("WorkingPath: " + work);
    conf.workingPath = work;
    conf.loadModel = load_model;
    conf.shards = shards;
    conf.shardsLrScale = shards_lrscale;
    LineConfiger ev = cf.getParamOptions("evaluation.setup");
    const std::string mode = lower(ev.getMainParam());
    const std::string testFile = mode == "test-set" ? ev.getString("-f") : "";
    LineConfiger ro = cf.getParamOptions("ratings.setup");
    if (!cf.contains("ratings.setup") || ro.getInt("-datatransformation", 1) > 0) {
        const int fmt = cmi_validate_data_format(ratingFile.c_str());
        if (fmt == 2 || fmt == 3)
            log(std::string("You rating data is in ") + (fmt == 2 ? "Loose" : "Compact") + " format. CARSKit is working on transformation on the data format...");
        int tree = 0;
        if (cmi_transform(ratingFile.c_str(), (work + "train.csv").c_str(), testFile.empty() ? nullptr : testFile.c_str(),
                          testFile.empty() ? nullptr : (work + "test.csv").c_str(), &tree) != CMI_OK)
            throw std::runtime_error("DataTransformer: " + std::string(cmi_dao_last_error(nullptr)));
    }
    Dao rateDao(work + "train.csv");
    log("Rating data set has been successfully loaded.");
    RatingData data = rateDao.ratingData();
    // runAlgorithm
    const std::string algoName = LineConfiger(cf.getString("recommender")).getMainParam();
    {   // the similarity-based CAMF recommenders are top-N models: their constructors switch the (static) isRankingPred on
        const std::string a = lower(algoName);
        if (a == "camf_ics" || a == "camf_lcs" || a == "camf_mcs") conf.isRankingPred = true;
    }
    log("With Setup: " + cf.getString("evaluation.setup"));
    std::vector<Measures> all;
    std::string name;
    std::mutex mu;
    std::string fold_error;
    int concurrent_folds = 1; // `cv -p on`: the folds that train side by side (set by the cv branch below)
    auto runFold = [&](const RatingData &tr, const RatingData &te, int fold, size_t slot) {
        try {
            Conf fc = conf; // fold -> GPU round robin (the reference runs one thread per fold, CARSKit.java:395-412)
            const int ngpu = cmi_device_count();
            if (ngpu > 1 && fold > 0) fc.device = (fold - 1) % ngpu;
            fc.deviceShare = std::max(1, (concurrent_folds + std::max(1, ngpu) - 1) / std::max(1, ngpu)); // the folds that share this fold's GPU
            auto algo = getRecommender(algoName, tr, te, fold, fc, log);
            const Measures m = algo->execute();
            std::lock_guard<std::mutex> g(mu);
            if (all.size() <= slot) all.resize(slot + 1);
            all[slot] = m;
            name = algo->algoName;
        } catch (const std::exception &e) { // the reference logs the exception of a fold thread (Recommender.java:1162-1171)
            std::lock_guard<std::mutex> g(mu);
            fold_error = e.what();
        }
    };
    if (mode == "cv") {
        int k = 0;
        const std::vector<int> labels = split_folds(data.n(), ev.getInt("-k", 5), conf.randSeed, &k);
        const bool parallel = ev.isOn("-p", true); // one thread per fold (CARSKit.java:395-412); each fold = its own cmi_handle/stream
        std::vector<std::thread> ts;
        std::vector<RatingData> trs((size_t)k), tes((size_t)k);
        for (int f = 1; f <= k; ++f) {
            std::vector<int64_t> tr, te;
            for (int64_t t = 0; t < data.n(); ++t) {
                if (data.r[(size_t)t] == 0.0) continue; // reshape() drops zero entries
                (labels[(size_t)t] == f ? te : tr).push_back(t);
            }
            trs[(size_t)f - 1] = data.subset(tr);
            tes[(size_t)f - 1] = data.subset(te);
        }
        concurrent_folds = parallel ? k : 1;
        for (int f = 1; f <= k; ++f) {
            if (parallel) ts.emplace_back(runFold, std::cref(trs[(size_t)f - 1]), std::cref(tes[(size_t)f - 1]), f, (size_t)f - 1);
            else runFold(trs[(size_t)f - 1], tes[(size_t)f - 1], f, (size_t)f - 1);
        }
        for (auto &t : ts) t.join();
    } else if (mode == "test-set") {
        Dao testDao(work + "test.csv", rateDao.h);
        RatingData test = testDao.ratingData();
        RatingData train = data; // id spaces of the union (the shared maps were extended by the test DAO)
        train.n_users = test.n_users;
        train.n_items = test.n_items;
        train.ctx_ptr = test.ctx_ptr;
        train.ctx_conds = test.ctx_conds;
        test.min_rate = data.min_rate;
        test.max_rate = data.max_rate;
        runFold(train, test, -1, 0);
    } else { // given-ratio: the reference draws Math.random() (unseedable); a seeded stream here
        const double ratio = ev.getDouble("-r", 0.8);
        JavaRandom rnd(conf.randSeed);
        std::vector<int64_t> tr, te;
        for (int64_t t = 0; t < data.n(); ++t) (rnd.nextDouble() < ratio ? tr : te).push_back(t);
        runFold(data.subset(tr), data.subset(te), -1, 0);
    }
    if (!fold_error.empty()) throw std::runtime_error(fold_error);
    Measures avg;
    for (const Measures &m : all)
        for (const auto &kv : m) avg[kv.first] += kv.second / (double)all.size();
    log("Final Results by " + name + ", " + evalInfo(avg, conf));
    if (precise && conf.isRankingPred)
        printf("PRECISE %s folds=%zu Pre10=%.17g Rec10=%.17g AUC10=%.17g MAP10=%.17g NDCG10=%.17g MRR10=%.17g\n", name.c_str(),
               all.size(), avg["Pre10"], avg["Rec10"], avg["AUC10"], avg["MAP10"], avg["NDCG10"], avg["MRR10"]);
    else if (precise) // machine-readable, full precision (for the parity tests)
        printf("PRECISE %s folds=%zu MAE=%.17g RMSE=%.17g\n", name.c_str(), all.size(), avg["MAE"], avg["RMSE"]);
    return 0;
}

int main(int argc, char **argv) {
    std::vector<std::string> configs;
    unsigned flags = 0;
    int iters = 0;
    bool precise = false, load_model = false;
    int shards = 1;
    double shards_lrscale = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-c") && i + 1 < argc) configs.push_back(argv[++i]);
        else if (!strcmp(argv[i], "--flags") && i + 1 < argc) flags = (unsigned)std::strtoul(argv[++i], nullptr, 0);
        else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = std::atoi(argv[++i]);
        else if (!strcmp(argv[i], "--precise")) precise = true;
        else if (!strcmp(argv[i], "--print-folds") && i + 3 < argc) {
            // test hook (no GPU, no config): the fold label of every matrix entry in CRS order, as DataSplitter.splitFolds assigns them
            int nf = 0;
            const std::vector<int> lab = carskit::split_folds(std::atoll(argv[i + 1]), std::atoi(argv[i + 2]), std::atoll(argv[i + 3]), &nf);
            printf("%d", nf);
            for (int v : lab) printf(" %d", v);
            printf("\n");
            return 0;
        } else if (!strcmp(argv[i], "--shards") && i + 1 < argc) shards = std::max(1, std::atoi(argv[++i])); // one recommender over N GPUs (cmi_group_*)
        else if (!strcmp(argv[i], "--shards-lrscale") && i + 1 < argc) shards_lrscale = std::atof(argv[++i]);
        else if (!strcmp(argv[i], "--load-model")) load_model = true; // evaluate the models `--save-model` left in the workspace instead of training
        else {
            fprintf(stderr, "usage: carskit-mi355x -c setting.conf [-c more.conf] [--flags N] [--iters N] [--precise] [--load-model] [--shards N [--shards-lrscale X]]\n");
            return 2;
        }
    }
    if (configs.empty()) configs.push_back("setting.conf");
    try {
        for (const std::string &c : configs) run(c, flags, iters, precise, load_model, shards, shards_lrscale);
    } catch (const std::exception &e) { // the reference logs e.getMessage() and a stack trace (CARSKit.java:96-101)
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
