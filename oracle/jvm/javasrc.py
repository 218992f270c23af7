"""A small interpreter for the subset of Java SOURCE the reference's hot path is written in -- TEST INFRASTRUCTURE, build container only.

The reference's recommenders exist as Java source only (src/carskit/**); there is no compiler and no JVM in this image, and no compiled
CARSKit classes ship with it.  oracle/jvm/interp.py executes the third-party jar the loops call into; this module
executes the loops themselves: it tokenises a .java file, finds method bodies, parses them (statements: blocks, local declarations,
if / for / for-each / while / switch / break / continue / return; expressions with Java's precedence: assignment and compound assignment,
?:, || &&, comparisons, + - * / %, unary, casts, ++ --, calls, field access, new) and evaluates them over
  * librec objects living in the bytecode VM (DenseMatrix, DenseVector, SparseMatrix, MatrixEntry, SymmMatrix: their methods run from
    the jar's bytecode),
  * host stand-ins for the JDK / guava classes (interp.py) and for the parts of CARSKit outside the loop (rateDao: id maps),
  * the fields of a `this` object the minting script fills in (hyper-parameters, model containers).
`this` has a CHAIN of source files (e.g. CAMF_CI.java -> CAMF.java -> ContextRecommender.java -> IterativeRecommender.java ->
Recommender.java): an unqualified call resolves to the first class in the chain that declares a method of that name and arity, `super.m()`
to the next one after the caller's -- Java's virtual dispatch for the methods on this path.

Numeric model: int -> Python int (32-bit wrap on + - *, truncating / and %), double -> Python float, float -> interp.JFloat (every
float-typed result rounded to binary32), boolean -> bool.  Math.pow(x, 2) is x * x (what fdlibm's and HotSpot's pow return for y == 2).
What is executed is the reference's text, read where it lies under /root/reference at minting time; nothing of it is copied into this
repository -- the fixtures hold numbers only (oracle/mint_reference_src.py).

Round 3 grew it from the SGD loops to everything either side of them (oracle/mint_reference_rank.py, oracle/mint_reference_dao.py,
oracle/check_java_binding.py): try / finally, array literals and casts, 2-D arrays, generic-method call syntax, overloads picked by
parameter type, objects whose class is itself interpreted (rateDao as DataDAO.java) and classes of static methods interpreted from source
on top of a jar class (carskit.eval.Measures over happy.coding.math.Measures), harness-provided native methods (NativeMF), and host
stand-ins written from the JDK / guava specifications: HashMap / HashSet / HashMultimap WITH java.util.HashMap's iteration order (JDK 8
bin table, resize at 0.75; tree bins refused), HashBiMap with its "value already present" refusal, Tree/LinkedHashMultimap, Multiset,
BufferedReader / Writer, String.split with Java's regex and limit rules, Integer.valueOf / Double.valueOf with their grammars.
"""
import math
import os
import re

from .interp import Box, JArray, JCollection, JFloat, JLong, JObject, JString, f32, i32
from .classfile import parse_descriptor

PRIMS = {"int", "double", "float", "boolean", "long", "char", "byte", "short"}
KEYWORDS = PRIMS | {"for", "if", "else", "while", "do", "switch", "case", "default", "break", "continue", "return", "new", "null", "true",
                    "false", "this", "super", "throw", "final", "instanceof", "try", "catch", "finally", "void"}

TOKEN_RE = re.compile(r"""
    (?P<ws>\s+) |
    (?P<lc>//[^\n]*) |
    (?P<bc>/\*.*?\*/) |
    (?P<num>(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fFdDlL]?) |
    (?P<str>"(?:\\.|[^"\\])*") |
    (?P<chr>'(?:\\.|[^'\\])') |
    (?P<id>[A-Za-z_$][A-Za-z_$0-9]*) |
    (?P<op>>>>=|<<=|>>=|>>>|\+\+|--|\+=|-=|\*=|/=|%=|&=|\|=|\^=|==|!=|<=|>=|&&|\|\||<<|->|[-+*/%=<>!?:;,.()\[\]{}@&|^~])
""", re.S | re.X)


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise SyntaxError("cannot tokenize at %r" % src[pos:pos + 30])
        pos = m.end()
        kind = m.lastgroup
        if kind in ("ws", "lc", "bc"):
            continue
        out.append((kind, m.group(kind)))
    return out


class Parser:
    def __init__(self, toks):
        self.t, self.p = toks, 0

    # -- token helpers
    def peek(self, k=0):
        return self.t[self.p + k] if self.p + k < len(self.t) else ("eof", "")

    def at(self, val, k=0):
        return self.peek(k)[1] == val and self.peek(k)[0] in ("op", "id")

    def eat(self, val=None):
        tok = self.peek()
        if val is not None and tok[1] != val:
            raise SyntaxError("expected %r, found %r (token %d)" % (val, tok[1], self.p))
        self.p += 1
        return tok

    # -- types: Name(.Name)* [<...>] ([])*
    def try_type(self):
        save = self.p
        tok = self.peek()
        if tok[0] != "id" or (tok[1] in KEYWORDS and tok[1] not in PRIMS):
            return None
        name = self.eat()[1]
        while self.at(".") and self.peek(1)[0] == "id":
            self.eat()
            name += "." + self.eat()[1]
        if self.at("<"):
            depth = 0
            while True:
                v = self.eat()[1]
                if v == "<":
                    depth += 1
                elif v == ">":
                    depth -= 1
                elif v == ">>":
                    depth -= 2
                elif v in (";", "(", ")", "{", "=") or self.peek()[0] == "eof":
                    self.p = save
                    return None
                if depth <= 0:
                    break
        dims = 0
        while self.at("[") and self.at("]", 1):
            self.eat(), self.eat()
            dims += 1
        return (name, dims)

    # -- statements
    def block(self):
        self.eat("{")
        body = []
        while not self.at("}"):
            body.append(self.statement())
        self.eat("}")
        return ("block", body)

    def statement(self):
        tok = self.peek()
        v = tok[1]
        if v == "{" and tok[0] == "op":
            return self.block()
        if v == ";" and tok[0] == "op":
            self.eat()
            return ("empty",)
        if tok[0] == "id":
            if v == "if":
                self.eat()
                self.eat("(")
                c = self.expr()
                self.eat(")")
                a = self.statement()
                b = None
                if self.at("else"):
                    self.eat()
                    b = self.statement()
                return ("if", c, a, b)
            if v == "for":
                return self.for_stmt()
            if v == "while":
                self.eat()
                self.eat("(")
                c = self.expr()
                self.eat(")")
                return ("while", c, self.statement())
            if v == "switch":
                return self.switch_stmt()
            if v == "break":
                self.eat(), self.eat(";")
                return ("break",)
            if v == "continue":
                self.eat(), self.eat(";")
                return ("continue",)
            if v == "return":
                self.eat()
                e = None if self.at(";") else self.expr()
                self.eat(";")
                return ("return", e)
            if v == "throw":
                self.eat()
                e = self.expr()
                self.eat(";")
                return ("throw", e)
            if v == "final":
                self.eat()
                return self.statement()
            if v == "assert":               # assertions are off in a JVM started without -ea: parsed, never evaluated
                self.eat()
                self.expr()
                if self.at(":"):
                    self.eat()
                    self.expr()
                self.eat(";")
                return ("block", [])
            if v == "try":               # catch clauses are parsed and ignored (a Java exception ends the run); finally IS executed
                self.eat()
                body = self.block()
                while self.at("catch"):
                    self.eat(), self.eat("(")
                    while not self.at(")"):
                        self.eat()
                    self.eat(")")
                    self.block()
                fin = None
                if self.at("finally"):
                    self.eat()
                    fin = self.block()
                return ("try", body, fin) if fin is not None else body
        decl = self.try_local_decl()
        if decl is not None:
            self.eat(";")
            return decl
        e = self.expr()
        self.eat(";")
        return ("expr", e)

    def try_local_decl(self):
        save = self.p
        ty = self.try_type()
        c_style = ty is not None and self.peek()[0] == "id" and self.peek(1)[1] == "[" and self.peek(2)[1] == "]"   # `String strs[] = ...`
        if ty is None or self.peek()[0] != "id" or self.peek()[1] in KEYWORDS or (self.peek(1)[1] not in ("=", ";", ",", ":") and not c_style):
            self.p = save
            return None
        decls = []
        while True:
            name = self.eat()[1]
            if self.at("[") and self.at("]", 1):
                self.eat(), self.eat()
            init = None
            if self.at("="):
                self.eat()
                if self.at("{"):            # T[] a = {x, y, ...};
                    self.eat()
                    items = []
                    while not self.at("}"):
                        items.append(self.expr_no_comma())
                        if self.at(","):
                            self.eat()
                    self.eat("}")
                    init = ("arraylit", ty, items)
                else:
                    init = self.expr_no_comma()
            decls.append((name, init))
            if self.at(","):
                self.eat()
                continue
            break
        return ("decl", ty, decls)

    def for_stmt(self):
        self.eat("for")
        self.eat("(")
        save = self.p
        ty = self.try_type()
        if ty is not None and self.peek()[0] == "id" and self.at(":", 1):
            name = self.eat()[1]
            self.eat(":")
            it = self.expr()
            self.eat(")")
            return ("foreach", ty, name, it, self.statement())
        self.p = save
        init = []
        if not self.at(";"):
            d = self.try_local_decl()
            if d is not None:
                init.append(d)
            else:
                init.append(("expr", self.expr_no_comma()))
                while self.at(","):
                    self.eat()
                    init.append(("expr", self.expr_no_comma()))
        self.eat(";")
        cond = None if self.at(";") else self.expr()
        self.eat(";")
        upd = []
        if not self.at(")"):
            upd.append(self.expr_no_comma())
            while self.at(","):
                self.eat()
                upd.append(self.expr_no_comma())
        self.eat(")")
        return ("for", init, cond, upd, self.statement())

    def switch_stmt(self):
        self.eat("switch")
        self.eat("(")
        e = self.expr()
        self.eat(")")
        self.eat("{")
        cases = []  # (label expr or None for default, [statements])
        while not self.at("}"):
            if self.at("case"):
                self.eat()
                lab = self.expr()
                self.eat(":")
                cases.append((lab, []))
            elif self.at("default"):
                self.eat(), self.eat(":")
                cases.append((None, []))
            else:
                cases[-1][1].append(self.statement())
        self.eat("}")
        return ("switch", e, cases)

    # -- expressions
    def expr(self):
        return self.expr_no_comma()

    def expr_no_comma(self):
        lhs = self.ternary()
        if self.peek()[0] == "op" and self.peek()[1] in ("=", "+=", "-=", "*=", "/=", "%="):
            op = self.eat()[1]
            rhs = self.expr_no_comma()
            return ("assign", op, lhs, rhs)
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.at("?"):
            self.eat()
            a = self.expr_no_comma()
            self.eat(":")
            b = self.expr_no_comma()
            return ("cond", c, a, b)
        return c

    LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">=", "instanceof"), ("<<", ">>", ">>>"), ("+", "-"),
              ("*", "/", "%")]

    def binary(self, lvl):
        if lvl == len(self.LEVELS):
            return self.unary()
        lhs = self.binary(lvl + 1)
        while self.peek()[1] in self.LEVELS[lvl] and self.peek()[0] in ("op", "id"):
            op = self.eat()[1]
            if op == "instanceof":
                self.try_type()
                lhs = ("lit", True)
                continue
            rhs = self.binary(lvl + 1)
            lhs = ("bin", op, lhs, rhs)
        return lhs

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("-", "+", "!", "~"):
            self.eat()
            return ("un", tok[1], self.unary())
        if tok[0] == "op" and tok[1] in ("++", "--"):
            self.eat()
            return ("preinc", tok[1], self.unary())
        if tok[0] == "op" and tok[1] == "(":  # a cast?
            save = self.p
            self.eat()
            ty = self.try_type()
            if ty is not None and self.at(")"):
                self.eat()
                nxt = self.peek()
                prim = ty[0] in PRIMS and ty[1] == 0
                starts_operand = nxt[0] in ("id", "num", "str", "chr") or (nxt[0] == "op" and nxt[1] in ("(", "!", "~"))
                if prim and nxt[0] == "op" and nxt[1] in ("-", "+"):
                    starts_operand = True
                if (prim or nxt[0] != "op" or nxt[1] in ("(", "!", "~")) and starts_operand and not (nxt[0] == "id" and nxt[1] == "instanceof"):
                    return ("cast", ty, self.unary())
            self.p = save
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.at("."):
                self.eat()
                if self.at("<"):            # explicit type arguments of a generic method call: HashMultimap.<Integer, Integer>create()
                    depth = 0
                    while True:
                        t = self.eat()[1]
                        depth += {"<": 1, ">": -1, ">>": -2}.get(t, 0)
                        if depth == 0:
                            break
                name = self.eat()[1]
                if self.at("("):
                    e = ("call", e, name, self.args())
                else:
                    e = ("field", e, name)
            elif self.at("["):
                self.eat()
                i = self.expr()
                self.eat("]")
                e = ("index", e, i)
            elif self.peek()[0] == "op" and self.peek()[1] in ("++", "--"):
                e = ("postinc", self.eat()[1], e)
            else:
                return e

    def args(self):
        self.eat("(")
        out = []
        while not self.at(")"):
            out.append(self.expr_no_comma())
            if self.at(","):
                self.eat()
        self.eat(")")
        return out

    def primary(self):
        kind, v = self.peek()
        if kind == "num":
            self.eat()
            s = v.rstrip("fFdDlL")
            if v[-1] in "fF":
                return ("lit", f32(float(s)))
            if v[-1] in "lL":
                return ("lit", JLong(int(s)))
            if v[-1] in "dD" or any(c in s for c in ".eE"):
                return ("lit", float(s))
            return ("lit", int(s))
        if kind == "str":
            self.eat()
            return ("lit", bytes(v[1:-1], "utf-8").decode("unicode_escape"))
        if kind == "chr":
            self.eat()
            return ("lit", JChar(ord(bytes(v[1:-1], "utf-8").decode("unicode_escape"))))
        if kind == "op" and v == "(":
            self.eat()
            e = self.expr()
            self.eat(")")
            return ("paren", e)
        if kind == "id":
            if v in ("true", "false"):
                self.eat()
                return ("lit", v == "true")
            if v == "null":
                self.eat()
                return ("lit", None)
            if v == "new":
                self.eat()
                ty = self.try_type_for_new()
                if self.at("[") and self.at("]", 1):      # new T[] { a, b, ... } / new T[][] { a, b }
                    while self.at("[") and self.at("]", 1):
                        self.eat(), self.eat()
                    self.eat("{")
                    items = []
                    while not self.at("}"):
                        items.append(self.expr_no_comma())
                        if self.at(","):
                            self.eat()
                    self.eat("}")
                    return ("arraylit", ty, items)
                if self.at("["):
                    self.eat()
                    n = self.expr()
                    self.eat("]")
                    if self.at("["):                      # new T[n][m] / new T[n][]
                        self.eat()
                        m = None if self.at("]") else self.expr()
                        self.eat("]")
                        return ("newarray2", ty, n, m)
                    return ("newarray", ty, n)
                a = self.args()
                if self.at("{"):  # anonymous class body: not on this path
                    raise SyntaxError("anonymous classes are not supported")
                return ("new", ty, a)
            self.eat()
            if self.at("("):
                return ("call", None, v, self.args())
            return ("name", v)
        raise SyntaxError("unexpected token %r" % (v,))

    def try_type_for_new(self):
        name = self.eat()[1]
        while self.at(".") and self.peek(1)[0] == "id":
            self.eat()
            name += "." + self.eat()[1]
        if self.at("<"):
            depth = 0
            while True:
                v = self.eat()[1]
                depth += {"<": 1, ">": -1, ">>": -2}.get(v, 0)
                if depth <= 0:
                    break
        return name


def find_methods(src):
    """{name: [(param names, body AST, class simple name)]} of the top-level class of a .java file"""
    toks = tokenize(src)
    methods, depth, i, cls = {}, 0, 0, None
    while i < len(toks):
        kind, v = toks[i]
        if kind == "op" and v == "@":          # annotation: skip its name (and arguments)
            i += 2
            if i < len(toks) and toks[i][1] == "(":
                d = 0
                while True:
                    d += {"(": 1, ")": -1}.get(toks[i][1], 0)
                    i += 1
                    if d == 0:
                        break
            continue
        if kind == "id" and v in ("class", "interface", "enum") and depth == 0 and cls is None:
            cls = toks[i + 1][1]
        if kind == "op" and v == "{":
            depth += 1
        elif kind == "op" and v == "}":
            depth -= 1
        elif depth == 1 and kind == "id" and v not in KEYWORDS and i + 1 < len(toks) and toks[i + 1][1] == "(" and toks[i - 1][0] in ("id", "op") \
                and (toks[i - 1][0] == "id" or toks[i - 1][1] in ("]", ">")) and toks[i - 1][1] not in ("new", "return", "=", ".") and v != cls:
            # a method declaration: Type name ( params ) [throws ...] { body }
            j = i + 2
            params, cur = [], []
            d = 1
            angle = 0          # commas inside a generic parameter type (Multimap<String, String> conditions) do not separate parameters
            while d > 0:
                t = toks[j]
                if t[1] == "(":
                    d += 1
                elif t[1] == ")":
                    d -= 1
                    if d == 0:
                        break
                elif t[1] == "<":
                    angle += 1
                elif t[1] == ">":
                    angle -= 1
                elif t[1] == ">>":
                    angle -= 2
                if t[1] == "," and d == 1 and angle == 0:
                    params.append(cur)
                    cur = []
                else:
                    cur.append(t)
                j += 1
            if cur:
                params.append(cur)
            j += 1
            while toks[j][1] not in ("{", ";"):
                j += 1
            if toks[j][1] == "{":
                # the body is only PARSED when the method is first called (a class has many methods this subset cannot parse and never
                # needs to); here it is skipped by matching braces
                d2, e = 0, j
                while True:
                    d2 += {"{": 1, "}": -1}.get(toks[e][1], 0) if toks[e][0] == "op" else 0
                    e += 1
                    if d2 == 0:
                        break
                names = [pp[-1][1] for pp in params]
                types = [" ".join(t[1] for t in pp[:-1] if t[1] != "final") for pp in params]
                methods.setdefault(v, []).append([names, types, ("lazy", toks, j), cls])
                i = e
                continue
        i += 1
    return methods, cls


class Break(Exception):
    pass


class Continue(Exception):
    pass


class Return(Exception):
    def __init__(self, v):
        self.v = v


class JavaExit(Exception):
    pass


class EnumConst:
    def __init__(self, cls, name):
        self.cls, self.name = cls, name

    def __repr__(self):
        return "%s.%s" % (self.cls, self.name)

    def __eq__(self, o):
        return isinstance(o, EnumConst) and o.name == self.name

    def __hash__(self):
        return hash(self.name)


def unbox(v):
    if isinstance(v, Box):
        return v.v if v.kind == "Double" else int(v.v)
    return v


class This:
    """An instance of a reference class: a chain of parsed source files (most derived first), fields, and host-provided members."""

    def __init__(self, vm, sources, class_map):
        self.vm = vm
        self.chain = []
        for path in sources:
            methods, cls = find_methods(open(path).read())
            self.chain.append((cls, methods))
        self.fields = {}
        self.host = {}          # name -> python callable(*args): methods of `this` outside the interpreted sources
        self.override = {}      # name -> python callable(*args) that REPLACES a source method (logging / debug output only)
        self.hooks = {}         # name -> python callable(this, args): runs before the source method of that name
        self.class_map = class_map   # simple class name -> jar class (for static calls / new), or a This of static methods
        self.static_super = None     # jar class whose static methods an interpreted class of statics inherits
        self.statements = 0

    def find(self, name, nargs, after=None, args=None):
        start = 0
        if after is not None:
            start = [c for c, _ in self.chain].index(after) + 1
        for cls, methods in self.chain[start:]:
            cands = [rec for rec in methods.get(name, []) if len(rec[0]) == nargs]
            if len(cands) > 1 and args is not None:      # overloads of one arity: the declared parameter types decide
                def fits(rec):
                    score = 0
                    for t, a in zip(rec[1], args):
                        t, a = t.split()[-1] if t else "", unbox(a)
                        if t == "String":
                            score += 2 if isinstance(a, str) else -9
                        elif t in ("int", "Integer"):
                            score += 2 if isinstance(a, int) and not isinstance(a, bool) else -9
                        elif t in ("double", "Double"):
                            score += 2 if isinstance(a, float) else (1 if isinstance(a, int) and not isinstance(a, bool) else -9)
                        elif t == "boolean":
                            score += 2 if isinstance(a, bool) else -9
                        elif isinstance(a, (str, int, float, bool)):
                            score -= 9
                    return score
                cands.sort(key=fits, reverse=True)
            for rec in cands[:1]:
                if rec[2][0] == "lazy":
                    p = Parser(rec[2][1])
                    p.p = rec[2][2]
                    rec[2] = p.block()
                return cls, rec[0], rec[1], rec[2]
        return None

    def call(self, name, args, after=None):
        if name in self.override and after is None:
            return self.override[name](*args)
        m = self.find(name, len(args), after, args)
        if m is None:
            if name in self.host:
                return self.host[name](*args)
            if self.static_super is not None:
                return vm_call(self.vm, None, self.static_super, name, args, static=True)
            raise KeyError("no source or host method %s/%d on %s" % (name, len(args), self.chain[0][0]))
        if name in self.hooks and after is None:
            self.hooks[name](self, args)
        cls, names, types, body = m
        env = Env(self, cls)
        for n, t, a in zip(names, types, args):
            env.declare(n, coerce(t.split()[-1] if t else "", a))
        try:
            env.exec(body)
        except Return as r:
            return r.v
        return None


def coerce(ty, v):
    """Java assignment conversion of v to declared type `ty` (simple name)"""
    v = unbox(v)
    if ty == "double" and isinstance(v, (int, float)) and not isinstance(v, bool):
        return float(v)
    if ty == "float" and isinstance(v, (int, float)) and not isinstance(v, bool):
        return f32(float(v))
    if ty == "int" and isinstance(v, float):
        raise TypeError("possible lossy conversion from double to int")
    return v


class Env:
    def __init__(self, this, cls):
        self.this, self.cls = this, cls
        self.scopes = [{}]
        self.types = [{}]

    def declare(self, name, v, ty=""):
        self.scopes[-1][name] = v
        self.types[-1][name] = ty

    def lookup_scope(self, name):
        for s, t in zip(reversed(self.scopes), reversed(self.types)):
            if name in s:
                return s, t
        return None, None

    # ---- statements
    def exec(self, node):
        self.this.statements += 1
        k = node[0]
        if k == "try":
            try:
                return self.exec(node[1])
            finally:
                self.exec(node[2])
        if k == "block":
            self.scopes.append({})
            self.types.append({})
            try:
                for s in node[1]:
                    self.exec(s)
            finally:
                self.scopes.pop()
                self.types.pop()
        elif k == "decl":
            ty = node[1][0] if node[1][1] == 0 else node[1][0] + "[]"
            for name, init in node[2]:
                v = coerce(ty, self.eval(init)) if init is not None else {"int": 0, "double": 0.0, "float": f32(0.0), "boolean": False}.get(ty)
                self.declare(name, v, ty)
        elif k == "expr":
            self.eval(node[1])
        elif k == "if":
            if self.truth(self.eval(node[1])):
                self.exec(node[2])
            elif node[3] is not None:
                self.exec(node[3])
        elif k == "for":
            self.scopes.append({})
            self.types.append({})
            try:
                for s in node[1]:
                    self.exec(s)
                while node[2] is None or self.truth(self.eval(node[2])):
                    try:
                        self.exec(node[4])
                    except Continue:
                        pass
                    except Break:
                        break
                    for u in node[3]:
                        self.eval(u)
            finally:
                self.scopes.pop()
                self.types.pop()
        elif k == "foreach":
            ty = node[1][0]
            for v in self.iterate(self.eval(node[3])):
                self.scopes.append({})
                self.types.append({})
                try:
                    self.declare(node[2], coerce(ty, v) if ty in PRIMS else from_host(v), ty)
                    self.exec(node[4])
                except Continue:
                    pass
                except Break:
                    break
                finally:
                    self.scopes.pop()
                    self.types.pop()
        elif k == "while":
            while self.truth(self.eval(node[1])):
                try:
                    self.exec(node[2])
                except Continue:
                    continue
                except Break:
                    break
        elif k == "switch":
            v = self.eval(node[1])
            matched = False
            try:
                for lab, body in node[2]:
                    if not matched:
                        if lab is None:
                            matched = True
                        else:
                            lv = EnumConst("?", lab[1]) if lab[0] == "name" and isinstance(v, EnumConst) else self.eval(lab)
                            matched = lv == v
                    if matched:
                        for s in body:
                            self.exec(s)
            except Break:
                pass
        elif k == "break":
            raise Break()
        elif k == "continue":
            raise Continue()
        elif k == "return":
            raise Return(None if node[1] is None else self.eval(node[1]))
        elif k == "throw":
            raise RuntimeError("Java throw: %r" % (self.eval(node[1]),))
        elif k == "empty":
            pass
        else:
            raise NotImplementedError(k)

    def truth(self, v):
        v = unbox(v)
        if isinstance(v, bool):
            return v
        if isinstance(v, int):
            return v != 0
        raise TypeError("not a boolean: %r" % (v,))

    def iterate(self, coll):
        if isinstance(coll, list):
            return list(coll)
        if isinstance(coll, JArray):
            return list(coll.data)
        vm = self.this.vm
        it = vm.invoke("interface", "java/lang/Iterable", "iterator", "()Ljava/util/Iterator;", [coll]) if isinstance(coll, JObject) else coll.jcall(vm, "iterator", "", [])

        def gen():
            while vm.invoke("interface", "java/util/Iterator", "hasNext", "()Z", [it]):
                yield vm.invoke("interface", "java/util/Iterator", "next", "()Ljava/lang/Object;", [it])
        return gen()

    # ---- expressions
    def eval(self, n):
        k = n[0]
        if k == "lit":
            return n[1]
        if k == "paren":
            return self.eval(n[1])
        if k == "name":
            return self.get_name(n[1])
        if k == "bin":
            return self.binop(n[1], n[2], n[3])
        if k == "un":
            v = unbox(self.eval(n[2]))
            if n[1] == "-":
                return f32(-v) if isinstance(v, JFloat) else (-v if isinstance(v, float) else i32(-v))
            if n[1] == "+":
                return v
            if n[1] == "!":
                return not self.truth(v)
            return i32(~v)
        if k == "cast":
            if n[1][1] > 0:            # a cast to an array type ((int[]) t[0]): a reference conversion, the value is unchanged
                return self.eval(n[2])
            v = unbox(self.eval(n[2]))
            ty = n[1][0]
            if ty == "float":
                return f32(float(v))
            if ty == "double":
                return float(v)
            if ty == "int":
                if isinstance(v, float):
                    return 0 if v != v else max(-2 ** 31, min(2 ** 31 - 1, int(v)))
                return i32(int(v))
            if ty == "long":
                return JLong(int(v))
            return v
        if k == "cond":
            return self.eval(n[2]) if self.truth(self.eval(n[1])) else self.eval(n[3])
        if k == "assign":
            return self.assign(n[1], n[2], n[3])
        if k in ("postinc", "preinc"):
            old = unbox(self.eval(n[2]))
            new = old + (1 if n[1] == "++" else -1)
            new = i32(new) if isinstance(old, int) else new
            self.store(n[2], new)
            return old if k == "postinc" else new
        if k == "field":
            return self.get_field(n[1], n[2])
        if k == "index":
            a, i = self.eval(n[1]), unbox(self.eval(n[2]))
            return a[i] if isinstance(a, list) else a.data[i]
        if k == "call":
            return self.call(n[1], n[2], [self.eval(a) for a in n[3]])
        if k == "new":
            return self.new(n[1], [self.eval(a) for a in n[2]])
        if k == "arraylit":
            return [self.eval(x) for x in n[2]]
        if k == "newarray2":
            zero = {"double": 0.0, "int": 0, "float": f32(0.0), "boolean": False, "long": JLong(0)}.get(n[1])
            rows = unbox(self.eval(n[2]))
            cols = None if n[3] is None else unbox(self.eval(n[3]))
            return [None if cols is None else [zero] * cols for _ in range(rows)]
        if k == "newarray":
            return [{"double": 0.0, "int": 0, "float": f32(0.0), "boolean": False, "long": JLong(0)}.get(n[1])] * unbox(self.eval(n[2]))
        raise NotImplementedError(k)

    def get_name(self, name):
        s, _ = self.lookup_scope(name)
        if s is not None:
            return s[name]
        if name in self.this.fields:
            return self.this.fields[name]
        if name == "this":
            return self.this
        return ("class", name)            # a class name used as a qualifier (Math, DenseMatrix, Measure, ...)

    def get_field(self, obj_node, name):
        obj = self.eval(obj_node)
        if isinstance(obj, tuple) and obj[0] == "class":
            if obj[1] in self.this.fields.get("__enums__", ()):
                return EnumConst(obj[1], name)
            if (obj[1], name) in STATIC_FIELDS:
                return STATIC_FIELDS[(obj[1], name)]
            holder = self.this.class_map.get(obj[1])
            if holder is not None and hasattr(holder, "consts") and name in holder.consts:
                return holder.consts[name]
            return EnumConst(obj[1], name)
        if obj is self.this:
            return self.this.fields[name]
        if isinstance(obj, list) and name == "length":
            return len(obj)
        if isinstance(obj, JArray) and name == "length":
            return len(obj.data)
        if isinstance(obj, JObject):
            return obj.fields[name]
        raise KeyError("field %s of %r" % (name, obj))

    def store(self, target, v):
        if target[0] == "paren":
            return self.store(target[1], v)
        if target[0] == "name":
            s, t = self.lookup_scope(target[1])
            if s is not None:
                s[target[1]] = coerce(t.get(target[1], ""), v)
                return
            if target[1] in self.this.fields:
                old = self.this.fields[target[1]]
                self.this.fields[target[1]] = float(v) if isinstance(old, float) and not isinstance(old, JFloat) and isinstance(v, int) \
                    and not isinstance(v, bool) else (f32(float(v)) if isinstance(old, JFloat) else v)
                return
            raise KeyError("assignment to unknown name %s" % target[1])
        if target[0] == "field":
            obj = self.eval(target[1])
            if obj is self.this:
                self.this.fields[target[2]] = v
            else:
                obj.fields[target[2]] = v
            return
        if target[0] == "index":
            a, i = self.eval(target[1]), unbox(self.eval(target[2]))
            v = unbox(v) if isinstance(v, Box) else v       # int[] / double[] elements: auto-unboxing on the way in
            if isinstance(a, list):
                a[i] = v
            else:
                a.data[i] = v
            return
        raise NotImplementedError("assignment target %s" % target[0])

    def assign(self, op, lhs, rhs):
        r = self.eval(rhs)
        if op != "=":
            cur = self.eval(lhs)
            r2 = self.arith(op[:-1], cur, r)
            # compound assignment casts back to the type of the left-hand side
            cu = unbox(cur)
            if isinstance(cu, JFloat):
                r2 = f32(float(r2))
            elif isinstance(cu, int) and not isinstance(cu, bool) and isinstance(r2, float):
                r2 = max(-2 ** 31, min(2 ** 31 - 1, int(r2)))
            r = r2
        self.store(lhs, unbox(r) if op != "=" else r)
        return r

    def binop(self, op, a, b):
        if op == "&&":
            return self.truth(self.eval(a)) and self.truth(self.eval(b))
        if op == "||":
            return self.truth(self.eval(a)) or self.truth(self.eval(b))
        x, y = self.eval(a), self.eval(b)
        if op in ("==", "!="):
            xu, yu = unbox(x), unbox(y)
            if isinstance(xu, (int, float, bool)) and isinstance(yu, (int, float, bool)):
                eq = xu == yu
            elif isinstance(xu, EnumConst) or isinstance(yu, EnumConst):
                eq = xu == yu
            else:
                eq = xu is yu
            return eq if op == "==" else not eq
        if op in ("<", ">", "<=", ">="):
            x, y = unbox(x), unbox(y)
            return {"<": x < y, ">": x > y, "<=": x <= y, ">=": x >= y}[op]
        return self.arith(op, x, y)

    def arith(self, op, x, y):
        x, y = unbox(x), unbox(y)
        if op == "+" and (isinstance(x, str) or isinstance(y, str)):
            return java_str(x) + java_str(y)
        if isinstance(x, bool) or isinstance(y, bool):
            if op in ("&", "|", "^"):
                return {"&": x and y, "|": x or y, "^": x != y}[op]
            raise TypeError("arithmetic on boolean")
        fl = isinstance(x, float) or isinstance(y, float)
        both_f32 = fl and all(isinstance(v, (JFloat, int)) for v in (x, y)) and any(isinstance(v, JFloat) for v in (x, y))
        if fl:
            x, y = float(x), float(y)
            if op == "+":
                r = x + y
            elif op == "-":
                r = x - y
            elif op == "*":
                r = x * y
            elif op == "/":
                r = (math.nan if (x == 0.0 or x != x) else math.copysign(math.inf, x) * math.copysign(1.0, y)) if y == 0.0 else x / y
            elif op == "%":
                r = math.fmod(x, y) if y != 0.0 else math.nan
            else:
                raise TypeError("operator %s on floating point" % op)
            return f32(r) if both_f32 else r
        wrap = (lambda v: JLong(((v + 2 ** 63) % 2 ** 64) - 2 ** 63)) if isinstance(x, JLong) or isinstance(y, JLong) else i32
        if op == "+":
            return wrap(x + y)
        if op == "-":
            return wrap(x - y)
        if op == "*":
            return wrap(x * y)
        if op in ("/", "%"):
            if y == 0:
                raise ZeroDivisionError("/ by zero")
            q = abs(x) // abs(y) * (1 if (x < 0) == (y < 0) else -1)
            return wrap(q if op == "/" else x - q * y)
        if op == "<<":
            return wrap(x << (y & 31))
        if op == ">>":
            return wrap(x >> (y & 31))
        if op == ">>>":
            return wrap((x & 0xFFFFFFFF) >> (y & 31))
        if op == "&":
            return wrap(x & y)
        if op == "|":
            return wrap(x | y)
        if op == "^":
            return wrap(x ^ y)
        raise NotImplementedError(op)

    # ---- calls
    def call(self, target, name, args):
        this, vm = self.this, self.this.vm
        if target is None:
            return this.call(name, args)
        if target[0] == "name" and target[1] == "super":
            return this.call(name, args, after=self.cls)
        obj = self.eval(target)
        if obj is this:
            return this.call(name, args)
        if isinstance(obj, tuple) and obj[0] == "class":
            return self.static_call(obj[1], name, args)
        if obj is None:
            raise RuntimeError("NullPointerException: .%s() on null" % name)
        if isinstance(obj, This):       # another object whose class is interpreted from source (rateDao)
            return obj.call(name, args)
        if isinstance(obj, JObject):
            return vm_call(vm, obj, obj.cls_name, name, args, static=False)
        if isinstance(obj, str):
            return string_method(obj, name, args)
        if isinstance(obj, list):
            raise KeyError("method %s on array" % name)
        if isinstance(obj, Box):
            return obj.jcall(vm, name, "", [unbox(a) for a in args])
        if isinstance(obj, EnumConst):
            if name in ("toString", "name"):
                return obj.name
            raise KeyError("enum method " + name)
        if hasattr(obj, "jcall"):
            r = obj.jcall(vm, name, "", [to_host(a) for a in args])
            return from_host(r)
        if hasattr(obj, name):
            return getattr(obj, name)(*[unbox(a) for a in args])
        raise KeyError("cannot call %s on %r" % (name, obj))

    def static_call(self, cls, name, args):
        a = [unbox(x) for x in args]
        key = (cls, name)
        if key in STATIC_CALLS:
            return STATIC_CALLS[key](*a)
        jar_cls = self.this.class_map.get(cls)
        if hasattr(jar_cls, "jstatic"):  # a class whose static (native) methods are provided by the harness
            return jar_cls.jstatic(name, args)
        if isinstance(jar_cls, This):   # a class of static methods interpreted from source; what it inherits comes from its `static_super`
            if jar_cls.find(name, len(args)) is not None:
                return jar_cls.call(name, args)
            jar_cls = jar_cls.static_super
        if jar_cls is not None:
            return vm_call(self.this.vm, None, jar_cls, name, args, static=True)
        if cls in ("Logs",):
            return None
        raise KeyError("static %s.%s" % (cls, name))

    def new(self, ty, args):
        simple = ty.split(".")[-1]
        if simple in ("ArrayList", "LinkedList"):
            if args and hasattr(args[0], "items"):
                return JCollection(list(args[0].items))
            return JCollection()
        if simple == "StringBuilder":
            return JStringBuilder(args[0] if args and isinstance(args[0], str) else "")
        if simple == "HashMap":
            return JHashMap()
        if simple == "HashSet":
            return JHashSet()
        if simple == "SimpleImmutableEntry":
            from .interp import HostEntry
            return HostEntry(to_host(args[0]), to_host(args[1]))
        if simple == "LinkedHashMap":
            return JMap()
        jar_cls = self.this.class_map.get(simple)
        if jar_cls is None:
            raise KeyError("new %s" % ty)
        o = self.this.vm.new_object(jar_cls)
        vm_call(self.this.vm, o, jar_cls, "<init>", args, static=False)
        return o


class JMap:
    """java.util.Map as the evaluated sources use it (put / get / containsKey / size); keys: enum constants, boxes, strings"""
    JAVA_TYPES = ("java/util/Map",)

    def __init__(self):
        self.d = {}

    def jcall(self, vm, name, desc, args):
        if name == "put":
            old = self.d.get(args[0])
            self.d[args[0]] = args[1]
            return old
        if name == "get":
            return self.d.get(args[0])
        if name == "containsKey":
            return args[0] in self.d
        if name == "size":
            return len(self.d)
        if name == "keySet":
            return JCollection(list(self.d.keys()))
        if name == "values":
            return JCollection(list(self.d.values()))
        if name == "inverse":
            m = JMap()
            m.d = {v: k for k, v in self.d.items()}
            return m
        raise KeyError("Map." + name)


def _java_hash(k):
    """Object.hashCode() of the key kinds the evaluated sources put into a java.util.HashMap"""
    if isinstance(k, JString):
        return string_method(k.s, "hashCode", []) & 0xFFFFFFFF
    if isinstance(k, Box) and k.kind == "Integer":
        return int(k.v) & 0xFFFFFFFF
    if isinstance(k, Box) and k.kind == "Double":
        import struct
        b = struct.unpack("<Q", struct.pack("<d", float(k.v)))[0]
        return (b ^ (b >> 32)) & 0xFFFFFFFF
    if isinstance(k, EnumConst):
        return hash(k.name) & 0xFFFFFFFF        # identity hash in Java: iteration order over enum keys is not defined, and not used
    raise KeyError("hashCode of %r" % (k,))


class JHashMap(JMap):
    """java.util.HashMap (JDK 8+) with its ITERATION ORDER: keySet() / values() walk the bin table by index, a bin in insertion order.
    The table starts at 16 bins and doubles whenever size exceeds 0.75 x capacity (resize splits a bin preserving order, so the order is a
    function of the final capacity and the insertion sequence).  The JDK itself is not part of the reference tree, so this is a SIMULATION
    of its documented implementation; bins that would have been treeified (>= 8 entries at capacity >= 64) are refused."""

    def _ordered(self):
        return _hash_order(list(self.d))

    def jcall(self, vm, name, desc, args):
        if name == "keySet":
            return JCollection(self._ordered())
        if name == "values":
            return JCollection([self.d[k] for k in self._ordered()])
        return super().jcall(vm, name, desc, args)


def _hash_order(keys):
    """iteration order of a java.util.HashMap / HashSet holding `keys` (given in insertion order): see JHashMap"""
    cap = 16
    while len(keys) > cap * 3 // 4:
        cap *= 2
    bins = {}
    for k in keys:
        h = _java_hash(k)
        bins.setdefault((h ^ (h >> 16)) & (cap - 1), []).append(k)
    if any(len(b) >= 8 for b in bins.values()):
        raise RuntimeError("a HashMap bin of 8+ entries: tree bins are not simulated")
    return [k for idx in sorted(bins) for k in bins[idx]]


class JHashSet(JCollection):
    """java.util.HashSet: add() ignores duplicates; iteration in HashMap order.  `items` is kept in that order after every change so the
    for-each of the evaluator and the iterator() of jar bytecode see it.  (A set that shrinks keeps its table in Java; the sets the
    evaluated code removes from -- candItems with -numIgnore -- are refused below once that would matter.)"""

    def __init__(self, items=None):
        super().__init__()
        self.inserted, self.peak = [], 0
        for x in items or []:
            self.jcall(None, "add", "", [x])

    def _reorder(self):
        self.peak = max(self.peak, len(self.inserted))
        cap, pcap = 16, 16
        while len(self.inserted) > cap * 3 // 4:
            cap *= 2
        while self.peak > pcap * 3 // 4:
            pcap *= 2
        if cap != pcap:
            raise RuntimeError("HashSet shrank below a resize boundary: its iteration order depends on the table it grew to")
        self.items = _hash_order(self.inserted)

    def jcall(self, vm, name, desc, args):
        if name == "add":
            if args[0] in self.inserted:
                return 0
            self.inserted.append(args[0])
            self._reorder()
            return 1
        if name == "remove":
            if args[0] in self.inserted:
                self.inserted.remove(args[0])
                self._reorder()
                return 1
            return 0
        return super().jcall(vm, name, desc, args)


class JHashMultimap:
    """guava HashMultimap<K, V> = HashMap<K, HashSet<V>>: keySet() and get(k) iterate in java.util.HashMap order"""
    JAVA_TYPES = ("com/google/common/collect/Multimap",)

    def __init__(self):
        self.d = JHashMap()

    def put(self, k, v):
        return self.jcall(None, "put", "", [Box(k, "Integer"), Box(v, "Integer")])

    def jcall(self, vm, name, desc, args):
        if name == "put":
            s = self.d.d.get(args[0])
            if s is None:
                s = self.d.d[args[0]] = JHashSet()
            return s.jcall(vm, "add", "", [args[1]])
        if name == "get":
            return self.d.d.get(args[0]) or JHashSet()
        if name == "containsKey":
            return args[0] in self.d.d
        if name == "keySet":
            return JCollection(self.d._ordered())
        if name == "size":
            return sum(len(v.items) for v in self.d.d.values())
        raise KeyError("HashMultimap." + name)


class JSetMultimap:
    """guava TreeMultimap (sorted=True: keys and each key's values in natural order) or LinkedHashMultimap (insertion order)"""

    def __init__(self, sorted_):
        self.sorted, self.d = sorted_, {}

    def _vals(self, k):
        v = self.d.get(k, [])
        return sorted(v, key=lambda x: x.s) if self.sorted else list(v)

    def jcall(self, vm, name, desc, args):
        if name == "put":
            v = self.d.setdefault(args[0], [])
            if args[1] in v:
                return False
            v.append(args[1])
            return True
        if name == "keySet":
            ks = list(self.d)
            return JCollection(sorted(ks, key=lambda x: x.s) if self.sorted else ks)
        if name == "get":
            return JCollection(self._vals(args[0]))
        if name == "size":
            return sum(len(v) for v in self.d.values())
        raise KeyError("Multimap." + name)


class JWriter:
    """java.io.BufferedWriter over a file: write / flush / close"""

    def __init__(self, path):
        self.fh = open(path, "w", newline="")

    def jcall(self, vm, name, desc, args):
        if name == "write":
            self.fh.write(from_host(args[0]))
            return None
        if name == "flush":
            self.fh.flush()
            return None
        if name == "close":
            self.fh.close()
            return None
        raise KeyError("BufferedWriter." + name)


def _copy_file(a, b):
    import shutil
    shutil.copyfile(a, b)


class JBiMap(JMap):
    """guava HashBiMap: put(k, v) with v already bound to another key throws (BiMap.put's contract)"""

    def jcall(self, vm, name, desc, args):
        if name == "put":
            for k2, v2 in self.d.items():
                if v2 == args[1] and k2 != args[0]:
                    raise RuntimeError("IllegalArgumentException: value already present: %r" % (args[1],))
        return super().jcall(vm, name, desc, args)


def _java_round(x):
    """Math.round(double): floor(x + 0.5) as a long (NaN -> 0)"""
    if x != x:
        return JLong(0)
    return JLong(max(-2 ** 63, min(2 ** 63 - 1, math.floor(x + 0.5))))


def to_host(v):
    """a Java value on its way into a host collection: primitives are boxed"""
    if isinstance(v, bool) or v is None or isinstance(v, (Box, JObject)):
        return v
    if isinstance(v, float):
        return Box(float(v), "Double")
    if isinstance(v, int):
        return Box(int(v), "Integer")
    if isinstance(v, str):
        return JString(v)
    return v


def from_host(v):
    if isinstance(v, JString):
        return v.s
    return v


class JChar(int):
    """a char value: an int in arithmetic (the results are plain ints), its character in a string concatenation"""


def java_str(v):
    if isinstance(v, str):
        return v
    if isinstance(v, JChar):
        return chr(v)
    if v is None:
        return "null"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, float):
        return repr(float(v))
    return str(v)


def string_method(s, name, args):
    a = [unbox(x) for x in args]
    if name == "split":
        # String.split(regex[, limit]): a regex; limit 0 (default) drops trailing empty strings, a negative limit keeps them
        limit = a[1] if len(a) > 1 else 0
        parts = re.split(a[0], s)
        if limit == 0:
            while len(parts) > 1 and parts[-1] == "":
                parts.pop()
            if parts == [""] and s != "":
                parts = []
        return parts
    if name == "trim":
        return s.strip("".join(chr(c) for c in range(0x21)))      # String.trim(): code points <= U+0020
    if name == "equals":
        return s == a[0]
    if name == "endsWith":
        return s.endswith(a[0])
    if name == "startsWith":
        return s.startswith(a[0])
    if name == "contains":
        return a[0] in s
    if name == "indexOf":
        return s.find(a[0])
    if name == "length":
        return len(s)
    if name == "toLowerCase":
        return s.lower()
    if name == "isEmpty":
        return s == ""
    if name == "hashCode":
        h = 0
        for ch in s:
            h = i32(31 * h + ord(ch))
        return h
    raise KeyError("String." + name)


class JStringBuilder:
    def __init__(self, s=""):
        self.s = s

    def jcall(self, vm, name, desc, args):
        if name == "append":
            self.s += java_str(unbox(from_host(args[0])))
            return self
        if name == "length":
            return len(self.s)
        if name == "toString":
            return JString(self.s)
        raise KeyError("StringBuilder." + name)


class JMultiset:
    """guava Multiset<Double> (HashMultiset): add / size / elementSet"""

    def __init__(self):
        self.items = []

    def jcall(self, vm, name, desc, args):
        if name == "add":
            self.items.append(args[0])
            return True
        if name == "size":
            return len(self.items)
        if name == "elementSet":
            seen, out = set(), []
            for x in self.items:
                if x not in seen:
                    seen.add(x)
                    out.append(x)
            return JCollection(out)
        raise KeyError("Multiset." + name)


class JReader:
    """java.io.BufferedReader over a text file: readLine() without the line terminator, null at the end"""

    def __init__(self, path):
        self.lines = open(path, newline="").read().split("\n")
        if self.lines and self.lines[-1] == "":
            self.lines.pop()
        self.lines = [ln[:-1] if ln.endswith("\r") else ln for ln in self.lines]
        self.i = 0

    def jcall(self, vm, name, desc, args):
        if name == "readLine":
            if self.i >= len(self.lines):
                return None
            self.i += 1
            return JString(self.lines[self.i - 1])
        if name == "close":
            return None
        raise KeyError("BufferedReader." + name)


def _parse_int(x):
    if isinstance(x, str):
        if not re.fullmatch(r"[+-]?\d+", x):
            raise RuntimeError("NumberFormatException: For input string: %r" % x)
        return int(x)
    return int(x)


_JAVA_DOUBLE = re.compile(r"[+-]?(NaN|Infinity|((\d+\.?\d*|\.\d+)([eE][+-]?\d+)?)[fFdD]?)")


def _parse_double(x):
    """Double.valueOf(String) (FloatingDecimal.readJavaFormatString): whitespace trimmed, decimal literal with an optional f/F/d/D
    suffix, NaN / Infinity; anything else is a NumberFormatException.  (Hex floating literals are not handled here.)"""
    if isinstance(x, str):
        t = x.strip("".join(chr(c) for c in range(0x21)))
        if not _JAVA_DOUBLE.fullmatch(t):
            raise RuntimeError("NumberFormatException: For input string: %r" % x)
        if t[-1] in "fFdD":
            t = t[:-1]
        return float(t.replace("Infinity", "inf").replace("NaN", "nan"))
    return float(x)


def _math_pow(x, y):
    if y == 2:
        return float(x) * float(x)      # fdlibm e_pow.c: "y is 2" -> x*x; HotSpot's intrinsic does the same
    return math.pow(x, y)


SYSTEM_PROPERTIES = {}   # -Dname=value of the evaluated run (strings), set by the harness


STATIC_CALLS = {
    ("Math", "pow"): _math_pow,
    ("Math", "abs"): lambda x: abs(x),
    ("Math", "sqrt"): lambda x: math.sqrt(x) if x >= 0 else math.nan,
    ("Math", "max"): lambda a, b: max(a, b),
    ("Math", "min"): lambda a, b: min(a, b),
    ("Math", "exp"): math.exp,
    ("Math", "round"): _java_round,
    ("Math", "log"): math.log,
    ("Double", "isNaN"): lambda x: x != x,
    ("Double", "isInfinite"): lambda x: math.isinf(x),
    ("Integer", "valueOf"): lambda x: Box(_parse_int(x), "Integer"),
    ("Integer", "parseInt"): lambda x: int(x),
    ("Double", "valueOf"): lambda x: Box(_parse_double(x), "Double"),
    ("String", "format"): lambda *a: "<formatted>",
    ("HashBasedTable", "create"): lambda: _guava_table(),
    ("HashMultimap", "create"): lambda: _guava_multimap(),
    ("String", "format"): lambda *a: "",
    ("TreeMultimap", "create"): lambda: JSetMultimap(True),
    ("LinkedHashMultimap", "create"): lambda: JSetMultimap(False),
    ("FileIO", "getWriter"): lambda path: JWriter(path),
    ("FileIO", "exist"): lambda path: os.path.exists(path),
    ("FileIO", "copyFile"): _copy_file,
    ("Integer", "parseInt"): lambda x: _parse_int(x),
    ("HashBiMap", "create"): lambda: JMap(),
    ("HashMultiset", "create"): lambda: JMultiset(),
    ("FileIO", "getReader"): lambda path: JReader(path),
    ("Strings", "last"): lambda s_, n_: s_[-n_:],
    ("Collections", "sort"): lambda coll: coll.items.sort(key=lambda b: b.v),
    ("Integer", "getInteger"): lambda name, dflt: Box(int(SYSTEM_PROPERTIES.get(name, dflt)), "Integer"),
    ("System", "getProperty"): lambda name, dflt=None: SYSTEM_PROPERTIES.get(name, dflt),
    ("Boolean", "getBoolean"): lambda name: str(SYSTEM_PROPERTIES.get(name, "false")).lower() == "true",
    ("Double", "parseDouble"): lambda x: _parse_double(x),
    ("Double", "toString"): lambda x: java_str(float(x)),
    ("Arrays", "asList"): lambda *a: JCollection([to_host(x) for x in a]),
    ("Logs", "debug"): lambda *a: None,
    ("Logs", "info"): lambda *a: None,
    ("Logs", "error"): lambda *a: None,
    ("Logs", "warn"): lambda *a: None,
}


def _guava_table():
    from .interp import GuavaTable
    return GuavaTable()


def _guava_multimap():
    return JHashMultimap()


def _exit(code):
    raise JavaExit("System.exit(%r)" % (code,))


STATIC_CALLS[("System", "exit")] = _exit
STATIC_FIELDS = {("Double", "MAX_VALUE"): 1.7976931348623157e308, ("Integer", "MAX_VALUE"): 2 ** 31 - 1, ("CARSKit", "isMeasuresOnly"): False}


def vm_call(vm, obj, cls, name, args, static):
    """call a jar method by NAME: pick the overload whose descriptor fits the argument values"""
    best = None
    c = cls
    while c and vm.jar.has(c):
        cf = vm.jar.load(c)
        for (nm, ds), m in cf.methods.items():
            if nm != name or m.code is None or m.static != static:
                continue
            ptypes, ret = parse_descriptor(ds)
            if len(ptypes) != len(args):
                continue
            conv, score, ok = [], 0, True
            for t, a in zip(ptypes, args):
                a = unbox(a) if t[0] not in "L[" else a
                if t == "I" and isinstance(a, int) and not isinstance(a, (bool, JLong)):
                    conv.append(a)
                    score += 2
                elif t == "D" and isinstance(a, float):
                    conv.append(float(a))
                    score += 2
                elif t == "D" and isinstance(a, int) and not isinstance(a, bool):
                    conv.append(float(a))
                    score += 1
                elif t == "Z" and isinstance(a, bool):
                    conv.append(int(a))
                    score += 2
                elif t == "J" and isinstance(a, int) and not isinstance(a, bool):
                    conv.append(JLong(a))
                    score += 1
                elif t[0] == "[" and isinstance(a, list):      # an array of the evaluator: the SAME list, so in-place changes are seen
                    conv.append(JArray(t[1:], a))
                    score += 1
                elif t[0] in "L[" and (a is None or isinstance(a, (JObject, JArray)) or hasattr(a, "jcall")):
                    conv.append(a)
                    score += 2 if isinstance(a, JObject) and ("L" + a.cls_name + ";") == t else 1
                elif t[0] == "L" and t in ("Ljava/lang/Object;", "Ljava/lang/Integer;", "Ljava/lang/Double;") and isinstance(a, (int, float)):
                    conv.append(to_host(a))
                    score += 1
                else:
                    ok = False
                    break
            if ok and (best is None or score > best[0]):
                best = (score, c, nm, ds, conv, ret)
        c = cf.super_name
    if best is None:
        raise KeyError("no overload of %s.%s for %r" % (cls, name, args))
    _, c, nm, ds, conv, ret = best
    vm.ensure_init(c)
    r = vm.invoke("static" if static else ("special" if name == "<init>" else "virtual"), c, nm, ds, ([] if static else [obj]) + conv)
    if ret == "Z":
        return bool(r)
    return from_host(r)
