"""Minimal Java class-file reader (JVMS chapter 4) -- TEST INFRASTRUCTURE, build container only.

Used by oracle/jvm/interp.py to EXECUTE methods of the reference's vendored third-party jar
(/root/reference/lib/librec-v1.4-alpha.jar: librec.data.DenseMatrix / DenseVector / SparseMatrix, librec.util.Randoms / Stats) so
that the L0 layer of the oracle (dot-product order, CRS iteration order, init streams, global mean) is pinned against the
reference's OWN bytecode rather than against a reading of it.  There is no JVM in this image; the jar never
leaves this container -- only the small input/output vectors minted from it are committed (tests/golden/librec_l0.json).
Nothing under carskit_amd/ imports this package.
"""
import struct
import zipfile

CONSTANT_Utf8, CONSTANT_Integer, CONSTANT_Float, CONSTANT_Long, CONSTANT_Double = 1, 3, 4, 5, 6
CONSTANT_Class, CONSTANT_String, CONSTANT_Fieldref, CONSTANT_Methodref, CONSTANT_InterfaceMethodref = 7, 8, 9, 10, 11
CONSTANT_NameAndType, CONSTANT_MethodHandle, CONSTANT_MethodType, CONSTANT_InvokeDynamic = 12, 15, 16, 18

ACC_STATIC, ACC_NATIVE, ACC_ABSTRACT = 0x0008, 0x0100, 0x0400


class Method:
    def __init__(self, cls, access, name, desc, code, max_locals, handlers):
        self.cls, self.access, self.name, self.desc = cls, access, name, desc
        self.code, self.max_locals, self.handlers = code, max_locals, handlers

    @property
    def static(self):
        return bool(self.access & ACC_STATIC)

    def __repr__(self):
        return "%s.%s%s" % (self.cls.name, self.name, self.desc)


class ClassFile:
    def __init__(self, data):
        self.d, self.p = data, 0
        if self.u4() != 0xCAFEBABE:
            raise ValueError("not a class file")
        self.minor, self.major = self.u2(), self.u2()
        n = self.u2()
        self.cp = [None] * n
        i = 1
        while i < n:
            tag = self.u1()
            if tag == CONSTANT_Utf8:
                ln = self.u2()
                self.cp[i] = (tag, self.d[self.p:self.p + ln].decode("utf-8", "replace"))
                self.p += ln
            elif tag == CONSTANT_Integer:
                self.cp[i] = (tag, struct.unpack(">i", self.take(4))[0])
            elif tag == CONSTANT_Float:
                self.cp[i] = (tag, struct.unpack(">f", self.take(4))[0])
            elif tag == CONSTANT_Long:
                self.cp[i] = (tag, struct.unpack(">q", self.take(8))[0])
                i += 1
            elif tag == CONSTANT_Double:
                self.cp[i] = (tag, struct.unpack(">d", self.take(8))[0])
                i += 1
            elif tag in (CONSTANT_Class, CONSTANT_String, CONSTANT_MethodType):
                self.cp[i] = (tag, self.u2())
            elif tag in (CONSTANT_Fieldref, CONSTANT_Methodref, CONSTANT_InterfaceMethodref, CONSTANT_NameAndType, CONSTANT_InvokeDynamic):
                self.cp[i] = (tag, self.u2(), self.u2())
            elif tag == CONSTANT_MethodHandle:
                self.cp[i] = (tag, self.u1(), self.u2())
            else:
                raise ValueError("constant pool tag %d" % tag)
            i += 1
        self.access = self.u2()
        self.name = self.class_name(self.u2())
        sup = self.u2()
        self.super_name = self.class_name(sup) if sup else None
        self.interfaces = [self.class_name(self.u2()) for _ in range(self.u2())]
        self.fields = []
        for _ in range(self.u2()):
            acc, nm, ds = self.u2(), self.utf8(self.u2()), self.utf8(self.u2())
            const = None
            for _ in range(self.u2()):
                an, ln = self.utf8(self.u2()), self.u4()
                body = self.take(ln)
                if an == "ConstantValue":
                    const = self.cp[struct.unpack(">H", body)[0]]
            self.fields.append((acc, nm, ds, const))
        self.methods = {}
        for _ in range(self.u2()):
            acc, nm, ds = self.u2(), self.utf8(self.u2()), self.utf8(self.u2())
            code, max_locals, handlers = None, 0, []
            for _ in range(self.u2()):
                an, ln = self.utf8(self.u2()), self.u4()
                body = self.take(ln)
                if an == "Code":
                    _, max_locals, cl = struct.unpack(">HHI", body[:8])
                    code = body[8:8 + cl]
                    q = 8 + cl
                    (ne,) = struct.unpack(">H", body[q:q + 2])
                    q += 2
                    for _ in range(ne):
                        s, e, h, ct = struct.unpack(">HHHH", body[q:q + 8])
                        handlers.append((s, e, h, self.class_name(ct) if ct else None))
                        q += 8
            self.methods[(nm, ds)] = Method(self, acc, nm, ds, code, max_locals, handlers)

    # -- byte cursor
    def take(self, n):
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def u1(self):
        return self.take(1)[0]

    def u2(self):
        return struct.unpack(">H", self.take(2))[0]

    def u4(self):
        return struct.unpack(">I", self.take(4))[0]

    # -- constant pool views
    def utf8(self, i):
        return self.cp[i][1]

    def class_name(self, i):
        return self.utf8(self.cp[i][1])

    def name_and_type(self, i):
        _, n, d = self.cp[i]
        return self.utf8(n), self.utf8(d)

    def member_ref(self, i):
        """(class name, member name, descriptor) of a Fieldref / Methodref / InterfaceMethodref"""
        _, c, nt = self.cp[i]
        return (self.class_name(c),) + self.name_and_type(nt)


class Jar:
    """one jar, or several searched in order (a class path)"""

    def __init__(self, path):
        paths = [path] if isinstance(path, str) else list(path)
        self.zips = [zipfile.ZipFile(q) for q in paths]
        self.where = {}
        for z in reversed(self.zips):
            for n in z.namelist():
                self.where[n] = z
        self.cache = {}

    def has(self, cls):
        return cls + ".class" in self.where

    def load(self, cls):
        if cls not in self.cache:
            self.cache[cls] = ClassFile(self.where[cls + ".class"].read(cls + ".class"))
        return self.cache[cls]


def parse_descriptor(desc):
    """'(IDLjava/lang/String;[D)V' -> (['I','D','Ljava/lang/String;','[D'], 'V')"""
    assert desc[0] == "("
    i, args = 1, []
    while desc[i] != ")":
        j = i
        while desc[j] == "[":
            j += 1
        if desc[j] == "L":
            j = desc.index(";", j)
        args.append(desc[i:j + 1])
        i = j + 1
    return args, desc[i + 1:]
