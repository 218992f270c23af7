"""The Java side of the drop-in boundary cannot be compiled here (no JDK, no jni.h -- SURVEY 8c), so it is checked AS TEXT:
  * every `public static native` method of java/carskit/alg/gpu/NativeMF.java has exactly one
    Java_carskit_alg_gpu_NativeMF_<name> definition in jni/carskit_jni.cpp with the matching JNI return and parameter types,
    and vice versa;
  * every cmi_* function the shim calls is declared in include/carskit_mi355x.h;
  * every NativeMF.<member> the Java sources use exists (method or constant), and the model / state / flag constants mirror the header;
  * every `recommender=` name of the hot path (CARSKit.java:461,463,700-707,742) has a *_GPU class, and INTEGRATION.md shows
    only factory cases, classes and natives that exist;
  * the shim holds no JNI critical region."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA_DIR = os.path.join(ROOT, "java", "carskit", "alg", "gpu")
JTYPE = {"void": "void", "int": "jint", "long": "jlong", "double": "jdouble", "boolean": "jboolean", "String": "jstring",
         "int[]": "jintArray", "double[]": "jdoubleArray", "double[][]": "jobjectArray", "long[]": "jlongArray"}


def _read(*parts):
    return open(os.path.join(ROOT, *parts)).read()


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def java_natives():
    src = _strip_comments(_read("java", "carskit", "alg", "gpu", "NativeMF.java"))
    out = {}
    for ret, name, params in re.findall(r"public\s+static\s+native\s+([\w\[\]]+)\s+(\w+)\s*\(([^)]*)\)\s*;", src):
        types = []
        for p in [q.strip() for q in params.split(",") if q.strip()]:
            t, _ = p.rsplit(None, 1)
            types.append(JTYPE[t.replace(" ", "")])
        assert name not in out, "overloaded native %s: JNI name mangling would differ" % name
        out[name] = (JTYPE[ret], types)
    return out


def jni_definitions():
    src = _strip_comments(_read("jni", "carskit_jni.cpp"))
    out = {}
    for ret, name, params in re.findall(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+Java_carskit_alg_gpu_NativeMF_(\w+)\s*\(([^)]*)\)", src):
        ps = [q.strip() for q in params.split(",")]
        assert re.match(r"JNIEnv\s*\*", ps[0]) and ps[1].split()[0] == "jclass", name   # static natives: (JNIEnv *, jclass, ...)
        assert name not in out, name
        out[name] = (ret, [p.split()[0] for p in ps[2:]])
    return out


def test_natives_and_jni_definitions_match_one_to_one():
    j, c = java_natives(), jni_definitions()
    assert len(j) >= 25
    assert set(j) == set(c), set(j) ^ set(c)
    for name in j:
        assert j[name] == c[name], (name, j[name], c[name])


def test_shim_calls_only_declared_abi_functions_and_is_logic_free():
    hdr = _read("include", "carskit_mi355x.h")
    declared = set(re.findall(r"\b(cmi_[a-z_0-9]+)\s*\(", hdr))
    shim = _strip_comments(_read("jni", "carskit_jni.cpp"))
    called = set(re.findall(r"\b(cmi_[a-z_0-9]+)\s*\(", shim))
    assert called and called <= declared, called - declared
    assert "Critical" not in shim              # no Get/ReleasePrimitiveArrayCritical around library calls
    assert not re.search(r"\bhip[A-Z]\w*\s*\(", shim)   # no device code or HIP runtime in the shim


def test_java_sources_use_only_existing_native_members_and_header_constants():
    natives = java_natives()
    nm = _strip_comments(_read("java", "carskit", "alg", "gpu", "NativeMF.java"))
    consts = {}
    for decl in re.findall(r"public\s+static\s+final\s+int\s+([^;]+);", nm):
        for part in decl.split(","):
            k, v = part.split("=")
            consts[k.strip()] = int(v.strip(), 0)
    hdr = _read("include", "carskit_mi355x.h")
    hconst = {k: int(v, 0) for k, v in re.findall(r"#define\s+(CMI_[A-Z_0-9]+)\s+\(?(-?(?:0x)?[0-9a-fA-F]+)u?\)?", hdr)}
    for jname, hname in [("BIASEDMF", "CMI_MODEL_BIASEDMF"), ("CAMF_C", "CMI_MODEL_CAMF_C"), ("CAMF_CI", "CMI_MODEL_CAMF_CI"),
                         ("CAMF_CU", "CMI_MODEL_CAMF_CU"), ("CAMF_CUCI", "CMI_MODEL_CAMF_CUCI"), ("PMF", "CMI_MODEL_PMF"),
                         ("P", "CMI_STATE_P"), ("Q", "CMI_STATE_Q"), ("USER_BIAS", "CMI_STATE_USER_BIAS"),
                         ("ITEM_BIAS", "CMI_STATE_ITEM_BIAS"), ("COND_BIAS", "CMI_STATE_COND_BIAS"), ("UC_BIAS", "CMI_STATE_UC_BIAS"),
                         ("IC_BIAS", "CMI_STATE_IC_BIAS"), ("FLAG_STATE_F64", "CMI_FLAG_STATE_F64"),
                         ("FLAG_SCHED_SERIAL", "CMI_FLAG_SCHED_SERIAL"), ("FLAG_STRICT", "CMI_FLAG_STRICT"),
                         ("FLAG_NO_GRAPH", "CMI_FLAG_NO_GRAPH"), ("FLAG_SCHED_CHAIN", "CMI_FLAG_SCHED_CHAIN"),
                         ("FLAG_NO_CHAIN", "CMI_FLAG_NO_CHAIN"), ("FLAG_SCHED_OWNER", "CMI_FLAG_SCHED_OWNER"),
                         ("FLAG_NO_OWNER", "CMI_FLAG_NO_OWNER"), ("FLAG_SPOKE_ARENA", "CMI_FLAG_SPOKE_ARENA"),
                         ("FLAG_NO_ARENA", "CMI_FLAG_NO_ARENA"), ("RANK_UCU", "CMI_RANK_UCU"), ("RANK_UC", "CMI_RANK_UC"),
                         ("SVDPP", "CMI_MODEL_SVDPP"), ("CAMF_ICS", "CMI_MODEL_CAMF_ICS"), ("CAMF_LCS", "CMI_MODEL_CAMF_LCS"),
                         ("CAMF_MCS", "CMI_MODEL_CAMF_MCS"), ("Y", "CMI_STATE_Y"), ("CC_MATRIX", "CMI_STATE_CC_MATRIX"),
                         ("CF_MATRIX", "CMI_STATE_CF_MATRIX"), ("C_VECTOR", "CMI_STATE_C_VECTOR")]:
        assert consts[jname] == hconst[hname], (jname, hname)
    used = set()
    for f in os.listdir(JAVA_DIR):
        if f.endswith(".java") and f != "NativeMF.java":
            used |= set(re.findall(r"NativeMF\.(\w+)", _strip_comments(open(os.path.join(JAVA_DIR, f)).read())))
    assert used and used <= set(natives) | set(consts), used - set(natives) - set(consts)
    # braces / parentheses balance in every source (the cheapest syntax check available without javac)
    for f in os.listdir(JAVA_DIR):
        src = _strip_comments(open(os.path.join(JAVA_DIR, f)).read())
        src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
        for a, b in ("{}", "()", "[]"):
            assert src.count(a) == src.count(b), (f, a)


FACTORY = {"biasedmf": "BiasedMF_GPU", "pmf": "PMF_GPU", "camf_c": "CAMF_C_GPU", "camf_ci": "CAMF_CI_GPU",
           "camf_cu": "CAMF_CU_GPU", "camf_cuci": "CAMF_CUCI_GPU", "fm": "FM_GPU",
           # SURVEY 8(f) N1 names (CARSKit.java:469,708-712)
           "svd++": "SVDPP_GPU", "camf_ics": "CAMF_ICS_GPU", "camf_lcs": "CAMF_LCS_GPU", "camf_mcs": "CAMF_MCS_GPU"}


def test_jni_shim_compiles_against_a_stub_jni_header():
    """`g++ -fsyntax-only` of the shim against tests/jni_stub/jni.h (the JNI types and the JNIEnv members the shim uses, signatures
    from the JNI specification) and the real include/carskit_mi355x.h: every statement type-checks against the C ABI.  A compile check,
    not parity evidence -- no JVM exists here."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the build image"
    res = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                          "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "jni", "carskit_jni.cpp")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_jni_shim_validates_array_lengths_before_every_tuple_or_csr_call():
    """mismatched Java arrays must throw IllegalArgumentException, not index native memory out of bounds."""
    shim = _strip_comments(_read("jni", "carskit_jni.cpp"))
    bodies = re.split(r"JNIEXPORT", shim)[1:]
    checked = 0
    for b in bodies:
        if re.search(r"\bexpand_pairs\s*\(", b) or "rp[r + 1]" in b:
            assert "bad_csr(" in b, b[:120]
            checked += 1
        if re.search(r"cmi_(group_|fm_)?(eval_ratings|set_eval_ratings|predict_batch|eval_rankings)\s*\(", b):
            assert "bad_tuples(" in b, b[:120]
            checked += 1
    assert checked >= 12
    assert "java/lang/IllegalArgumentException" in shim


def test_every_hot_path_recommender_name_has_a_class_and_integration_md_shows_only_what_exists():
    integ = _read("INTEGRATION.md")
    natives = java_natives()
    for name, cls in FACTORY.items():
        path = os.path.join(JAVA_DIR, cls + ".java")
        assert os.path.exists(path), cls
        src = _strip_comments(open(path).read())
        assert re.search(r"public\s+class\s+%s\s+extends\s+\w+" % cls, src)
        assert re.search(r"public\s+%s\s*\(\s*SparseMatrix\s+\w+\s*,\s*SparseMatrix\s+\w+\s*,\s*int\s+\w+\s*\)" % cls, src)  # the factory's constructor
        assert "buildModel()" in src
        assert re.search(r'case\s+"%s_gpu"\s*:\s*return\s+new\s+carskit\.alg\.gpu\.%s\(' % (re.escape(name), cls), integ), name
    for cls in re.findall(r"carskit\.alg\.gpu\.(\w+)\(", integ):
        assert os.path.exists(os.path.join(JAVA_DIR, cls + ".java")), cls
    for member in re.findall(r"NativeMF\.(\w+)\(", integ):
        assert member in natives, "INTEGRATION.md shows NativeMF.%s, which does not exist" % member
    hdr = _read("include", "carskit_mi355x.h")
    for fn in set(re.findall(r"`(cmi_[a-z_0-9]+)`", integ)) - {"cmi_handle", "cmi_fm_handle", "cmi_dao_handle", "cmi_group_handle"}:
        assert re.search(r"\b%s\s*\(" % fn, hdr), fn
