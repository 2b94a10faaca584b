"""The map-scale factorisation's broadcast FMAs are hand-written 64-bit DPP instructions (inline assembly, csrc/gba.hip: v_fmac_f64_dpp / v_mov_b64_dpp with
row_newbcast).  The hardware wants two wait states between a VALU write of a VGPR and a DPP read of it; a violation reads a stale value -- silently.  This compiles
gba.hip for gfx950 (no GPU needed) and walks the instruction stream of every kernel that carries DPP instructions (tools/isa_census.py --dpp-hazards)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_dpp_read_right_behind_a_valu_write():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_census.py"), "--dpp-hazards", os.path.join(ROOT, "eao_fusion_amd", "csrc", "gba.hip"), "k_bal"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "DPP instructions" in l]
    assert any(l.startswith("k_bal_step") for l in lines) and any(l.startswith("k_bal_diag") for l in lines), r.stdout
    for l in lines:
        assert l.rstrip().endswith(" 0 hazard violations"), l
        if l.startswith(("k_bal_step", "k_bal_diag0")):
            assert int(l.split()[1]) >= 600, l      # the broadcast FMAs are there (616 per 32 x 32 factorisation, 496 per row solve)
