"""Static check of the SHIPPED gfx950 code (no GPU needed): no buffer store of more than 64 bits of data is followed, inside the two wait
states the gfx940+ rule asks for, by a VALU write of its data registers.  The compiler exempts buffer stores whose soffset is an SGPR
from that rule; on gfx950 they are not exempt, and that exemption is what corrupted a record's low word now and then in round 5 (the
owner epoch beside another owner epoch; docs/history/r06.md 1, tools/micro/store_data_hazard.hip).  owner_st_words keeps the data
registers alive across an `s_nop 1`; this test fails if any translation unit of the library reintroduces the pattern."""
import os

from tools.exp import scan_store_data_hazard as scan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scanner_sees_the_pattern_and_the_fix():
    bad = """
0000000000001000 <kernel_a>:
	buffer_store_dwordx4 v[2:5], v76, s[52:55], s24 offen sc0 sc1
	v_mov_b32_e32 v3, s8
	v_mov_b32_e32 v2, v74
"""
    good = bad.replace("\tv_mov_b32_e32 v3, s8", "\ts_nop 1\n\tv_mov_b32_e32 v3, s8")
    other = bad.replace("v_mov_b32_e32 v3, s8", "v_mov_b32_e32 v9, s8").replace("v_mov_b32_e32 v2, v74", "v_mov_b32_e32 v7, v74")
    late = bad.replace("\tv_mov_b32_e32 v3, s8", "\tv_mov_b32_e32 v9, s8\n\tv_mov_b32_e32 v10, s8\n\tv_mov_b32_e32 v3, s8")
    assert [f[0] for f in scan.scan_disassembly(bad)] == ["kernel_a"]
    assert scan.scan_disassembly(good) == [] and scan.scan_disassembly(other) == [] and scan.scan_disassembly(late) == []
    small = bad.replace("dwordx4 v[2:5]", "dwordx2 v[2:3]")      # 64 bits of data: no hazard
    assert scan.scan_disassembly(small) == []


def test_shipped_library_has_no_exposed_wide_buffer_store():
    lib = os.path.join(ROOT, "carskit_amd", "lib", "libcarskit_mi355x.so")
    found, stores = scan.scan(lib)
    assert stores > 1000          # the owner kernels' record stores are there (the scan really read the code objects)
    assert found == [], "store-data hazard: %d wide buffer stores have their data registers rewritten too early, e.g. %s" % (len(found), found[:3])
