"""Static guard against the gfx9-family store-data hazard in the SHIPPED binary (host-only; used by tests/test_isa_hazards.py).

Rule (ISA manual, "manually inserted wait states"): a VMEM store of more than 64 bits of data needs 1 wait state (2 on gfx940+) before a
VALU instruction writes one of the VGPRs that hold its data -- the store reads them after it has issued.  The compiler inserts the wait
states itself, EXCEPT for buffer (MUBUF) stores whose soffset operand is an SGPR: LLVM's GCNHazardRecognizer::createsVALUHazard takes
the manual's word that those are exempt.  On gfx950 they are not (tools/micro/store_data_hazard.hip: 2.6 % of such stores picked up the
new value with 0 wait states, none with 1; docs/history/r06.md 1): that is what made the owner epoch's records carry a wrong low word
now and then in round 5.

This script disassembles every gfx950 code object inside a shared library and lists the buffer stores of more than 64 bits of data
(dwordx3 / dwordx4) that are followed, within the next WAIT issue slots and without an s_nop covering the distance, by a VALU write
of one of their data registers.  usage: python tools/exp/scan_store_data_hazard.py [lib.so]  (exit status 1 if any is found)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
WAIT = 2    # wait states the gfx940+ rule asks for


def code_objects(lib):
    """the gfx950 ELF images of every offload bundle in the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.devnull])
        blob = open(fat, "rb").read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tsz].decode()
            q += 24 + tsz
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def vregs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_disassembly(text):
    """[(kernel, store instruction, slots after it, overwriting instruction)]"""
    found, kern = [], "?"
    insts = []
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            insts.append(("label", m.group(1)))
            continue
        t = line.split("//")[0].strip()
        if t and not t.endswith(":") and re.match(r"^[a-z]", t):
            insts.append(("inst", t))
    for i, (kind, t) in enumerate(insts):
        if kind == "label":
            kern = t
            continue
        if not re.match(r"buffer_store_(dwordx[34]|format_xyzw?)\b", t):
            continue
        data = vregs(t.split(None, 1)[1].split(",")[0])
        slots = 0
        for kind2, u in insts[i + 1:i + 1 + 2 * WAIT]:
            if kind2 == "label" or slots >= WAIT:
                break
            m = re.match(r"s_nop (\d+)", u)
            if m:
                slots += int(m.group(1)) + 1
                continue
            slots += 1
            if u.startswith("v_") and not u.startswith("v_cmp") and not u.startswith("v_readlane") and not u.startswith("v_readfirstlane"):
                if vregs(u.split(None, 1)[1].split(",")[0]) & data:
                    found.append((kern, t, slots, u))
                    break
    return found


def scan(lib):
    found, stores = [], 0
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True,
                                  check=True).stdout
        stores += len(re.findall(r"buffer_store_dwordx[34]", text))
        found += scan_disassembly(text)
    return found, stores


if __name__ == "__main__":
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "carskit_amd", "lib", "libcarskit_mi355x.so")
    found, stores = scan(lib)
    for kern, st, slots, u in found[:40]:
        print("%s\n    %s\n    +%d: %s" % (kern, st, slots, u))
    print("%d buffer stores of more than 64 bits in %s; %d with a VALU write of their data registers inside %d wait states" % (stores, lib, len(found), WAIT))
    sys.exit(1 if found else 0)
