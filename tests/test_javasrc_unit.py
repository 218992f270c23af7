"""Unit tests of the Java-source interpreter (oracle/jvm/javasrc.py) on Java text written for the purpose: the semantics the minted goldens
rest on -- numeric promotion, 32-bit wrap-around, truncating division, compound-assignment narrowing, string concatenation, switch
fall-through, ternaries, for-each, try / finally, overloads -- and the JDK behaviours its host stand-ins restate (java.util.HashMap
iteration order, String.split, Integer / Double parsing).  Expected values are what the Java Language Specification / the JDK documentation
prescribe, worked out by hand."""
import os

import pytest

from oracle.jvm import javasrc
from oracle.jvm.interp import Box, JString

SRC = r'''
package t;
public class T {
    int wrap() { int x = 2147483647; x = x + 1; return x; }
    int mulwrap() { int x = 65536; return x * x + 7; }
    int idiv() { return (-7) / 2; }
    int imod() { return (-7) % 2; }
    double ddiv() { return 7 / 2; }
    double promote() { int size = 3; float reg = 0.1f; return size + reg; }
    double promote2() { float a = 0.1f; double b = a; return b; }
    int narrow() { int x = 10; x *= 0.35; return x; }
    int narrow2() { int x = 7; x += 1.9; return x; }
    String concat() { int a = 1; double d = 2; return "x" + a + d + 'c' + true + null; }
    String concat2() { int a = 1; int b = 2; return a + b + "s" + a + b; }
    int sw(int v) { int r = 0; switch (v) { case 1: r += 1; case 2: r += 10; break; case 3: r += 100; break; default: r = -1; } return r; }
    int tern(int v) { return v > 2 ? v < 5 ? 1 : 2 : 3; }
    int loop() { int s = 0; for (int i = 0; i < 10; ++i) { if (i == 3) continue; if (i == 7) break; s += i; } return s; }
    int foreach() { int[] a = new int[] {4, 5, 6}; int s = 0; for (int v : a) s = s * 10 + v; return s; }
    int fin() { int r = 1; try { r = 2; return r; } finally { f = 5; } }
    int f;
    int over(int a) { return 1; }
    int over(String a) { return 2; }
    int over(double a) { return 3; }
    int callover() { return over(1) * 100 + over("s") * 10 + over(1.5); }
    int shifts() { int x = -8; return (x >>> 28) * 100 + (x << 2); }
    boolean lazy() { int[] a = new int[1]; return a.length > 5 && a[9] == 0; }
    double twod() { double[][] m = new double[2][3]; m[1][2] = 4.5; double[][] r = new double[2][]; r[0] = m[1]; return r[0][2] + m[0].length; }
    int charmath() { String s = "abc"; return s.length() * 100 + s.indexOf("c") * 10 + (s.endsWith("bc") ? 1 : 0); }
}
'''


@pytest.fixture(scope="module")
def this(tmp_path_factory):
    p = tmp_path_factory.mktemp("j") / "T.java"
    p.write_text(SRC)
    t = javasrc.This(None, [str(p)], {})
    t.fields["f"] = 0
    return t


def test_integer_arithmetic_wraps_and_truncates(this):
    assert this.call("wrap", []) == -2147483648
    assert this.call("mulwrap", []) == 7                    # 2^32 wraps to 0
    assert this.call("idiv", []) == -3 and this.call("imod", []) == -1
    assert this.call("ddiv", []) == 3.0                     # integer division happens before the widening
    assert this.call("shifts", []) == 15 * 100 - 32          # >>> shifts zeros in (`>>` is not tokenised: it would clash with `List<List<X>>`)


def test_binary_numeric_promotion_int_plus_float_is_a_float_sum(this):
    import struct
    f = lambda x: struct.unpack("f", struct.pack("f", x))[0]
    assert this.call("promote", []) == f(3 + f(0.1))        # the FM `size + regLw` case: rounded to binary32, then widened
    assert this.call("promote", []) != 3 + f(0.1)
    assert this.call("promote2", []) == f(0.1)


def test_compound_assignment_narrows(this):
    assert this.call("narrow", []) == 3 and this.call("narrow2", []) == 8


def test_string_concatenation(this):
    assert this.call("concat", []) == "x12.0ctruenull"
    assert this.call("concat2", []) == "3s12"


def test_control_flow(this):
    assert [this.call("sw", [v]) for v in (1, 2, 3, 4)] == [11, 10, 100, -1]
    assert [this.call("tern", [v]) for v in (1, 3, 9)] == [3, 1, 2]
    assert this.call("loop", []) == 0 + 1 + 2 + 4 + 5 + 6
    assert this.call("foreach", []) == 456
    assert this.call("lazy", []) is False                   # && does not evaluate a[9]
    assert this.call("fin", []) == 2 and this.fields["f"] == 5


def test_overloads_arrays_strings(this):
    assert this.call("callover", []) == 123
    assert this.call("twod", []) == 7.5
    assert this.call("charmath", []) == 321


def test_hashmap_iteration_order_is_the_jdk8_bin_order():
    m = javasrc.JHashMap()
    for key in (17, 1, 33, 2):                              # 17, 1 and 33 share bin 1 of 16: insertion order inside the bin, then bin 2
        m.jcall(None, "put", "", [Box(key, "Integer"), Box(0, "Integer")])
    assert [k.v for k in m.jcall(None, "keySet", "", []).items] == [17, 1, 33, 2]
    for key in range(100, 109):                             # 13 entries: the table doubles to 32 bins and 17 leaves bin 1
        m.jcall(None, "put", "", [Box(key, "Integer"), Box(0, "Integer")])
    assert [k.v for k in m.jcall(None, "keySet", "", []).items] == [1, 33, 2, 100, 101, 102, 103, 104, 105, 106, 107, 108, 17]
    s = javasrc.JHashMap()
    for key in ("cherry", "apple", "banana"):               # String.hashCode, spread by h ^ (h >>> 16): bins 1, 1, 0 of 16
        s.jcall(None, "put", "", [JString(key), Box(0, "Integer")])
    assert [k.s for k in s.jcall(None, "keySet", "", []).items] == ["banana", "cherry", "apple"]
    assert javasrc.string_method("cherry", "hashCode", []) == -1361513063
    assert javasrc.string_method("banana", "hashCode", []) == -1396355227
    assert javasrc.string_method("apple", "hashCode", []) == 93029210


def test_string_split_and_number_parsing_follow_the_jdk():
    sp = lambda s, *a: javasrc.string_method(s, "split", list(a))
    assert sp("a,b,,", ",") == ["a", "b"]                   # limit 0 drops trailing empty strings
    assert sp("a,b,,", ",", -1) == ["a", "b", "", ""]
    assert sp(",a", ",") == ["", "a"]                       # a leading empty string stays
    assert sp("u1\t i2,3", "[\t,]+") == ["u1", " i2", "3"]
    assert sp("", ",") == [""]
    assert javasrc._parse_double(" 4.0d ") == 4.0 and javasrc._parse_double("1e0") == 1.0 and javasrc._parse_double(".5f") == 0.5
    for bad in ("", "4,0", "0x10", "four"):
        with pytest.raises(RuntimeError):
            javasrc._parse_double(bad)
    assert javasrc._parse_int("-12") == -12 and javasrc._parse_int("+7") == 7
    for bad in (" 1", "1 ", "1.0", ""):
        with pytest.raises(RuntimeError):
            javasrc._parse_int(bad)
    assert javasrc.string_method(" \tx\n", "trim", []) == "x"
