"""CPU restatement of the reference's data path for the binary / compact rating formats.  TEST INFRASTRUCTURE ONLY,
PARITY UNPINNED by the reference (no tests, no JVM); pinned by hand-derived expectations on the reference's own
sample file (sampleData/train_binary.csv, copied as data to tests/golden/) and by agreement with the product's
C++ implementation (carskit_amd/csrc/data_dao.cpp), which was written separately.

  read_data            DataDAO.readData                  src/carskit/data/processor/DataDAO.java:166-354
  JavaHashMap          java.util.HashMap (JDK 8+) as an explicit bin table with resize splitting -- a SIMULATION
                       of the published algorithm, deliberately not the closed form the product uses
  compact_to_binary    DataTransformer.TransformationFromCompactToBinary / PublishNewRatingFiles / getHeader
                       src/carskit/data/processor/DataTransformer.java:231-259, 266-329, 155-163
"""
import re


def jtrim(s):
    b, e = 0, len(s)
    while b < e and ord(s[b]) <= 0x20:
        b += 1
    while e > b and ord(s[e - 1]) <= 0x20:
        e -= 1
    return s[b:e]


def jsplit(s, regex, limit=0):
    """java.lang.String.split semantics (limit 0 drops trailing empties, -1 keeps them)."""
    parts, last = [], 0
    for m in re.finditer(regex, s):
        if m.end() == 0:      # zero-width/leading match at 0 yields no leading empty only for zero-width matches
            if m.start() == m.end():
                continue
        parts.append(s[last:m.start()])
        last = m.end()
    if not parts and last == 0:
        return [s]
    parts.append(s[last:])
    if limit == 0:
        while parts and parts[-1] == "":
            parts.pop()
    return parts


def read_lines(path):
    data = open(path, "rb").read().decode("latin-1")
    lines, cur, i, pending = [], [], 0, False
    while i < len(data):
        c = data[i]
        i += 1
        if c in "\r\n":
            if c == "\r" and i < len(data) and data[i] == "\n":
                i += 1
            lines.append("".join(cur))
            cur, pending = [], False
        else:
            cur.append(c)
            pending = True
    if pending:
        lines.append("".join(cur))
    return lines


def jdouble(s):
    """Double.valueOf(String): whitespace trimmed, an optional f/F/d/D suffix (DataDAO.java:228)"""
    t = jtrim(s)
    return float(t[:-1] if t[-1:] in "dDfF" else t)


def read_data(path):
    lines = read_lines(path)
    header = jsplit(jtrim(lines[0]), r"[\t,]+")
    dim_ids, cond_keys, cond_dim, empty = {}, [], [], []
    for i in range(3, len(header)):
        context = jtrim(header[i])
        dim = jtrim(jsplit(context, ":")[0]) if context != "" else ""
        dimc = dim_ids.setdefault(dim, len(dim_ids))
        cond_keys.append(context)
        cond_dim.append(dimc)
        if context.endswith(":na"):
            empty.append(i - 3)
    users, items, uis, ctxs = {}, {}, {}, {}
    ui_user, ui_item, ctx_conds = [], [], {}
    table, scale, n_lines = {}, set(), 0
    for line in lines[1:]:
        data = jsplit(jtrim(line), ",", -1)
        user, item, rate = data[0], data[1], jdouble(data[2])
        scale.add(rate)
        n_lines += 1
        row = users.setdefault(user, len(users))
        col = items.setdefault(item, len(items))
        key = "%d,%d" % (row, col)
        if key not in uis:
            uis[key] = len(uis)
            ui_user.append(row)
            ui_item.append(col)
        uic = uis[key]
        conds = [i - 3 for i in range(3, len(data)) if int(jtrim(data[i])) == 1]
        ctx = ",".join(str(c) for c in conds)
        cc = ctxs.setdefault(ctx, len(ctxs))
        ctx_conds[cc] = conds
        table[(uic, cc)] = rate
    entries = sorted(table.items())
    return {
        "users": list(users), "items": list(items), "uis": list(uis), "ctxs": list(ctxs), "dims": list(dim_ids),
        "conds": cond_keys, "cond_dim": cond_dim, "empty": empty, "ui_user": ui_user, "ui_item": ui_item,
        "ctx_conds": [ctx_conds[c] for c in range(len(ctxs))], "num_ratings": n_lines,
        "scale": sorted(scale), "ui": [k[0] for k, _ in entries], "ctx": [k[1] for k, _ in entries],
        "r": [v for _, v in entries],
    }


def jstring_hash(s):
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h


class JavaHashMap:
    """java.util.HashMap<String, V> insertion + iteration, as an explicit table of bins (lists).  put() appends to
    the bin (tail insertion), resize doubles the table and splits every bin into lo/hi preserving order; iteration
    walks the table in index order.  Treeification (bins of >= 8 entries) is not simulated; `max_bin` records the
    largest bin ever seen so a caller can tell whether it could have happened."""

    def __init__(self):
        self.table = None
        self.size = 0
        self.threshold = 0
        self.max_bin = 0

    @staticmethod
    def _spread(key):
        h = jstring_hash(key)
        return h ^ (h >> 16)

    def _resize(self):
        if self.table is None:
            self.table = [[] for _ in range(16)]
            self.threshold = 12
            return
        old = self.table
        ocap = len(old)
        new = [[] for _ in range(ocap * 2)]
        for idx, b in enumerate(old):
            for (k, v, h) in b:
                (new[idx] if (h & ocap) == 0 else new[idx + ocap]).append((k, v, h))
        self.table = new
        self.threshold = int(ocap * 2 * 0.75)

    def put(self, key, value):
        if self.table is None:
            self._resize()
        h = self._spread(key)
        b = self.table[h & (len(self.table) - 1)]
        for i, (k, _, hh) in enumerate(b):
            if hh == h and k == key:
                b[i] = (k, value, hh)
                return
        b.append((key, value, h))
        self.max_bin = max(self.max_bin, len(b))
        self.size += 1
        if self.size > self.threshold:
            self._resize()

    def keys(self):
        return [k for b in (self.table or []) for (k, _, _) in b]

    def items(self):
        return [(k, v) for b in (self.table or []) for (k, v, _) in b]


def compact_to_binary(in_path):
    """Returns the lines of the reference's rewritten train.csv (without line terminators)."""
    lines = read_lines(in_path)
    header = jsplit(lines[0], ",", -1)
    dims = [jtrim(h).lower() for h in header[3:]]
    conditions = {}           # LinkedHashMultimap: dim -> ordered set of conditions
    newlines = JavaHashMap()
    for line in lines[1:]:
        strs = jsplit(line, ",", -1)
        rc = {}
        for i in range(3, 3 + len(dims)):
            cond = jtrim(strs[i]).lower() or "na"
            rc[dims[i - 3]] = cond
            conds = conditions.setdefault(dims[i - 3], [])
            if cond not in conds:
                conds.append(cond)
        newlines.put(line, rc)
    out = ["User, Item, Rating" + "".join(", %s:%s" % (d, c) for d in conditions for c in conditions[d])]
    for key, rc in newlines.items():
        bits = []
        for d in conditions:
            for c in conditions[d]:
                bits.append("1" if rc[d] == c else "0")
        skey = jsplit(key, ",", -1)
        if len(skey) > 3:
            key = ",".join(jtrim(x).lower() for x in skey[:3])
        out.append(key + "," + ",".join(bits))
    return out, newlines.max_bin


# ---- the remaining DataTransformer paths (loose, binary->binary, merged conditions) and the shared-map test DAO -------

def validate_format(lines):
    """CARSKit.validateDataFormat (src/carskit/main/CARSKit.java:179-215); 0 where the reference throws (NullPointerException on a
    missing data line, ArrayIndexOutOfBounds on a one-column header or a short data line, NumberFormatException from Integer.valueOf)."""
    if len(lines) < 2:
        return 0
    sh, sd = jsplit(lines[0], ",", -1), jsplit(lines[1], ",", -1)
    if len(sh) < 2:
        return 0
    if jtrim(sh[-2]).lower() == "dimension" and jtrim(sh[-1]).lower() == "condition":
        return 2
    for i in range(3, len(sh)):
        if ":" not in sh[i]:
            return 3
        if i >= len(sd) or re.fullmatch(r"[+-]?\d+", sd[i]) is None or not -2 ** 31 <= int(sd[i]) < 2 ** 31:
            return 0
        v = int(sd[i])
        if v > 0 and any(ch > "1" for ch in str(v)):      # isBinaryNumber: Java's % keeps the sign, so a negative number always passes
            return 3
    return 1


class _Conds:
    """dim -> ordered conditions; sorted=True models guava's TreeMultimap, else LinkedHashMultimap."""

    def __init__(self, sorted_=False):
        self.sorted, self.d = sorted_, {}

    def put(self, dim, cond):
        v = self.d.setdefault(dim, [])
        if cond not in v:
            v.append(cond)

    def dims(self):
        return sorted(self.d) if self.sorted else list(self.d)

    def conds(self, dim):
        return sorted(self.d[dim]) if self.sorted else list(self.d[dim])

    def copy(self):
        c = _Conds(self.sorted)
        c.d = {k: list(v) for k, v in self.d.items()}
        return c


def _collect(lines, fmt, conds):
    header = jsplit(lines[0], ",", -1)
    if fmt == 1:
        for h in header[3:]:
            s = jsplit(h, ":", -1)
            conds.put(jtrim(s[0]).lower(), jtrim(s[1]).lower())
    elif fmt == 2:
        for line in lines[1:]:
            s = jsplit(line, ",", -1)
            conds.put(jtrim(s[3]).lower(), jtrim(s[4]).lower() or "na")
    else:
        for line in lines[1:]:
            s = jsplit(line, ",", -1)
            for i in range(3, len(header)):
                conds.put(jtrim(header[i]).lower(), jtrim(s[i]).lower() or "na")


def _transform_one(lines, fmt, is_test, given):
    conds = given.copy() if given is not None else _Conds(False)
    header = jsplit(lines[0], ",", -1)
    newlines = JavaHashMap()
    store = {}
    if fmt == 3:
        dims = [jtrim(h).lower() for h in header[3:]]
        for line in lines[1:]:
            s = jsplit(line, ",", -1)
            rc = {}
            for i in range(3, 3 + len(dims)):
                c = jtrim(s[i]).lower() or "na"
                rc[dims[i - 3]] = c
                if not is_test:
                    conds.put(dims[i - 3], c)
            newlines.put(line, rc)
    elif fmt == 2:
        for line in lines[1:]:
            s = jsplit(line, ",", -1)
            key = ",".join(jtrim(x).lower() for x in s[:3])
            c = jtrim(s[4]).lower() or "na"
            dim = jtrim(s[3]).lower()
            if not is_test:
                conds.put(dim, c)
            if key not in store:
                store[key] = {}
                newlines.put(key, store[key])
            store[key][dim] = c
    else:
        for line in lines[1:]:
            s = jsplit(line, ",", -1)
            rc = {}
            for i in range(3, len(header)):
                if int(jtrim(s[i]).lower()) == 0:
                    continue
                rs = jsplit(header[i], ":", -1)
                rc[jtrim(rs[0]).lower()] = jtrim(rs[1]).lower()
                if not is_test:
                    conds.put(jtrim(rs[0]).lower(), jtrim(rs[1]).lower())
            newlines.put(line, rc)
    out = ["User, Item, Rating" + "".join(", %s:%s" % (d, c) for d in conds.dims() for c in conds.conds(d))]
    loose = fmt == 2
    for key, rc in newlines.items():
        bits = []
        for d in conds.dims():
            dc = rc.get(d)
            if dc is None and not loose:
                raise TypeError("NullPointerException in the reference")
            na = dc is None or dc == "na"
            done = False
            for c in conds.conds(d):
                if loose:
                    if na:
                        hit = c == "na"
                    else:
                        hit = (not done) and c == dc
                    done = done or hit
                    bits.append("1" if hit else "0")
                else:
                    bits.append("1" if dc == c else "0")
        skey = jsplit(key, ",", -1)
        if len(skey) > 3:
            key = ",".join(jtrim(x).lower() for x in skey[:3])
        out.append(key + "," + ",".join(bits))
    return out, newlines.max_bin


def transform(train_path, test_path=None):
    """DataTransformer.run(): returns (train lines, test lines or None, largest HashMap bin seen)."""
    tr = read_lines(train_path)
    ftr = validate_format(tr)
    if test_path is None:
        if ftr == 1:
            return tr, None, 0
        out, mb = _transform_one(tr, ftr, False, None)
        return out, None, mb
    te = read_lines(test_path)
    fte = validate_format(te)
    merged = _Conds(True)
    _collect(tr, ftr, merged)
    _collect(te, fte, merged)
    for d in merged.dims():
        if "na" not in merged.d[d]:
            merged.put(d, "na")
    a, m1 = _transform_one(tr, ftr, False, merged)
    b, m2 = _transform_one(te, fte, True, merged)
    return a, b, max(m1, m2)


def read_data_shared(train_path, test_path):
    """Train DAO, then the test DAO built over the same (mutated) maps (CARSKit.java:335-340).  Returns the test
    matrix entries with the union id tables."""
    tr = read_data(train_path)
    users = {k: i for i, k in enumerate(tr["users"])}
    items = {k: i for i, k in enumerate(tr["items"])}
    uis = {k: i for i, k in enumerate(tr["uis"])}
    ctxs = {k: i for i, k in enumerate(tr["ctxs"])}
    ui_user, ui_item = list(tr["ui_user"]), list(tr["ui_item"])
    ctx_conds = {i: c for i, c in enumerate(tr["ctx_conds"])}
    lines = read_lines(test_path)
    table, n_lines = {}, 0
    for line in lines[1:]:
        data = jsplit(jtrim(line), ",", -1)
        rate = jdouble(data[2])
        n_lines += 1
        row = users.setdefault(data[0], len(users))
        col = items.setdefault(data[1], len(items))
        key = "%d,%d" % (row, col)
        if key not in uis:
            uis[key] = len(uis)
            ui_user.append(row)
            ui_item.append(col)
        conds = [i - 3 for i in range(3, len(data)) if int(jtrim(data[i])) == 1]
        ctx = ",".join(str(c) for c in conds)
        cc = ctxs.setdefault(ctx, len(ctxs))
        ctx_conds[cc] = conds
        table[(uis[key], cc)] = rate
    entries = sorted(table.items())
    return {"users": list(users), "items": list(items), "uis": list(uis), "ctxs": list(ctxs), "ui_user": ui_user,
            "ui_item": ui_item, "ctx_conds": [ctx_conds[c] for c in range(len(ctxs))], "num_ratings": n_lines,
            "ui": [k[0] for k, _ in entries], "ctx": [k[1] for k, _ in entries], "r": [v for _, v in entries]}
