"""setting.conf parsing with the reference's semantics: java.util.Properties file -> happy.coding.io.FileConfiger,
each value re-parsed by LineConfiger ("config / flags"):
  * split the value on [,\\t ]; the first token that does not start with '-' is the main parameter;
  * a token starting with '-' that is NOT numeric opens a new option key; every other token is appended to the
    current key's value list (so `-max -1` reads -1 as the value of -max);
  * isOn(v) <=> v in {on, true} (case-insensitive);
  * getPath(key): `key` if present, else `key.wins` on Windows, else `key.lins`.
Numbers that the reference parses as Java float are exposed as the double the float promotes to."""
import os
import re
import struct


def java_float(s):
    return struct.unpack("f", struct.pack("f", float(s)))[0]


def _is_numeric(tok):
    try:
        float(tok)
        return True
    except ValueError:
        return False


class LineConfiger:
    def __init__(self, line):
        self.main = None
        self.params = {}
        cur = None
        for tok in [t for t in re.split(r"[,\t ]", line.strip()) if t != ""]:
            if tok.startswith("-") and not _is_numeric(tok):
                cur = tok
                self.params.setdefault(cur, [])
            elif cur is None and self.main is None:
                self.main = tok
            elif cur is not None:
                self.params[cur].append(tok)
            elif self.main is not None:
                pass  # stray tokens before any option are ignored by the callers

    def get_main_param(self):
        return self.main

    def is_main_on(self):
        return (self.main or "").lower() in ("on", "true")

    def contains(self, key):
        return key in self.params

    def get_string(self, key, default=None):
        v = self.params.get(key)
        return v[0] if v else default

    def get_float(self, key, default=None):
        v = self.get_string(key)
        return java_float(v) if v is not None else default

    def get_double(self, key, default=None):
        v = self.get_string(key)
        return float(v) if v is not None else default

    def get_int(self, key, default=None):
        v = self.get_string(key)
        return int(v) if v is not None else default

    get_long = get_int

    def is_on(self, key, default=False):
        v = self.get_string(key)
        return default if v is None else v.lower() in ("on", "true")


def load_properties(path):
    """java.util.Properties.load for the subset setting.conf uses: key=value lines, # / ! comments, backslash escapes."""
    props = {}
    for raw in open(path, encoding="latin-1").read().splitlines():
        line = raw.lstrip()
        if not line or line[0] in "#!":
            continue
        m = re.match(r"((?:\\.|[^=:\s\\])*)\s*[=:\s]\s*(.*)", line)
        if not m:
            props[line] = ""
            continue
        unesc = lambda t: re.sub(r"\\(.)", lambda g: {"t": "\t", "n": "\n", "r": "\r", "f": "\f"}.get(g.group(1), g.group(1)), t)
        props[unesc(m.group(1))] = unesc(m.group(2))
    return props


class FileConfiger:
    def __init__(self, path):
        self.path = path
        self.props = load_properties(path)

    def contains(self, key):
        return key in self.props

    def get_string(self, key, default=None):
        v = self.props.get(key)
        return v.strip() if v is not None else default

    def get_int(self, key, default=None):
        v = self.get_string(key)
        return int(v) if v is not None else default

    def get_param_options(self, key):
        v = self.get_string(key)
        return LineConfiger(v) if v is not None else None

    def get_path(self, key):
        if key in self.props:
            return self.get_string(key)
        return self.get_string(key + (".wins" if os.name == "nt" else ".lins"))
