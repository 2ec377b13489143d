"""The C-ABI library loads and exports every symbol include/magnet_hip.h declares; the ctypes
mirror of the argument struct has the C layout; argument errors come back as codes (no compute
is launched here — that is what the -m gpu tests do)."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(REPO, "include", "magnet_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"MAGNET_API\s+[\w\s\*]+?\b(magnet_\w+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    assert {"magnet_cost_volume_cw", "magnet_pack_features", "magnet_gaussian_update",
            "magnet_upsample_depth", "magnet_version", "magnet_last_error"} <= set(syms)


def test_library_exports_every_declared_symbol(hip_lib):
    from magnet_amd import lib
    for s in _declared_symbols():
        assert hasattr(hip_lib, s), f"{s} declared in include/magnet_hip.h but not exported"
    assert set(_declared_symbols()) == set(lib.API_SYMBOLS)
    assert hip_lib.magnet_version() == 400


@pytest.mark.parametrize("struct", ["MagnetCostVolumeArgs", "MagnetConvArgs"])
def test_struct_layout_matches_c(hip_lib, struct):
    from magnet_amd import lib as L
    A = getattr(L, struct)
    fields = [f[0] for f in A._fields_]
    prog = '#include "%s"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%%zu", sizeof(%s));' % (HEADER, struct)
    for f in fields:
        prog += 'printf(" %%zu", offsetof(%s, %s));' % (struct, f)
    prog += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c"); exe = os.path.join(d, "t")
        open(c, "w").write(prog)
        subprocess.check_call(["gcc", c, "-o", exe])
        vals = [int(x) for x in subprocess.check_output([exe]).decode().split()]
    assert vals[0] == ctypes.sizeof(A)
    assert vals[1:] == [getattr(A, f).offset for f in fields]


def test_argument_errors_are_codes_not_crashes(hip_lib):
    from magnet_amd.lib import MagnetCostVolumeArgs
    assert hip_lib.magnet_cost_volume_cw(None, None) == 1                       # MAGNET_E_NULL
    assert b"NULL" in hip_lib.magnet_last_error()
    a = MagnetCostVolumeArgs()                                                  # all zero
    assert hip_lib.magnet_cost_volume_cw(ctypes.byref(a), None) == 1
    assert hip_lib.magnet_pack_features(None, None, 1, 8, 4, 4, 0, 0, None) == 1
    assert hip_lib.magnet_pack_features(16, 16, 1, 7, 4, 4, 0, 0, None) == 2       # MAGNET_E_DIM (F % 8)
    assert hip_lib.magnet_pack_features(16, 16, 1, 8, 4, 4, 9, 0, None) == 3       # MAGNET_E_DTYPE
    assert hip_lib.magnet_pack_features(16, 24, 1, 8, 4, 4, 0, 0, None) == 4
    assert hip_lib.magnet_pack_features(16, 16, 1, 8, 4, 4, 0, 2, None) == 2       # pad in {0,1}
    assert hip_lib.magnet_pack_gmm(None, 16, 1, 4, 4, None) == 1       # MAGNET_E_ALIGN
    assert hip_lib.magnet_gaussian_update(16, 16, 16, 0, 5, None) == 2
    assert hip_lib.magnet_upsample_depth(16, 16, 16, 1, 2, 4, 4, 3, None) == 2  # k must be 1,2,4,8
    # convolution / F-Net entry points
    from magnet_amd.lib import MagnetConvArgs, _conv_protos, _fnet_protos
    _conv_protos(hip_lib); _fnet_protos(hip_lib)
    c = MagnetConvArgs()
    assert hip_lib.magnet_conv_mfma(ctypes.byref(c), None) == 1
    c.in_hi = c.in_lo = c.w_hi = c.w_lo = c.bias = c.out_hi = c.out_lo = 16
    c.rows, c.cin, c.cout_pad, c.taps, c.wp = 128, 48, 128, 9, 10
    assert hip_lib.magnet_conv_mfma(ctypes.byref(c), None) == 2                 # cin % 32
    c.cin = 64; c.cout_pad = 48
    assert hip_lib.magnet_conv_mfma(ctypes.byref(c), None) == 2                 # unsupported width
    c.cout_pad = 64; c.taps = 5
    assert hip_lib.magnet_conv_mfma(ctypes.byref(c), None) == 2                 # taps in {1,4,9}
    c.taps = 9; c.repad = 1
    assert hip_lib.magnet_conv_mfma(ctypes.byref(c), None) == 2 and b"repad" in hip_lib.magnet_last_error()
    c.repad = 0; c.border_hp, c.border_pad = 7, 1                               # 128 rows are not a whole number of 7x10 grids
    assert hip_lib.magnet_conv_mfma(ctypes.byref(c), None) == 2
    c.border_hp = 0; c.tail_w_hi = 16
    assert hip_lib.magnet_conv_mfma(ctypes.byref(c), None) == 1                 # fused tail without its other pointers
    assert hip_lib.magnet_fnet_stem(None, 16, 16, 16, 16, 1, 8, 8, None) == 1
    assert hip_lib.magnet_fnet_stem(16, 16, 16, 16, 16, 1, 1, 8, None) == 2
    assert hip_lib.magnet_space_to_depth(16, 16, 16, 16, 1, 12, 4, 4, 2, None) == 2     # C % 8
    assert hip_lib.magnet_avgpool_cl(16, 16, 320, 1, 8, 8, 2, 16, 128, 16, 16, None) == 2   # window larger than the map
    assert hip_lib.magnet_upsample_bilinear_cl(16, 32, 1, 2, 32, 16, 24, 320, 1, 8, 8, 2, None) == 4   # out_lo misaligned
    assert hip_lib.magnet_cost_volume_f_backward(None, 16, 16, 16, None) == 1
    # D over the limit
    a.ref_feat_cl = a.src_feat_pad = a.src_gmm_pad = a.poses = a.is_valid = a.intM = a.rays = a.cost = 16
    a.d_volume = 16
    a.B = a.V = a.h = a.w = 1; a.F = 8; a.D = 257
    assert hip_lib.magnet_cost_volume_cw(ctypes.byref(a), None) == 2
    assert b"MAGNET_MAX_CANDIDATES" in hip_lib.magnet_last_error()


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under magnet_amd/ may import, load or link it."""
    pkg = os.path.join(REPO, "magnet_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                for needle in ("libmagnet_oracle", "import oracle", "from oracle", "oracle.oracle",
                               "magnet_oracle_", "cost_volume_oracle.h"):
                    assert needle not in txt, f"{f} references the oracle ({needle})"


def test_host_raises_without_gpu_tensors(hip_lib):
    import torch
    from magnet_amd import lib
    x = torch.zeros(1, 8, 4, 4)
    with pytest.raises(lib.MagnetError, match="no CPU fallback"):
        lib.pack_features(x)
    with pytest.raises(lib.MagnetError, match="no CPU fallback"):
        lib.gaussian_update(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4))


def test_graft_entry_version_check_follows_the_header():
    """__graft_entry__.build() compares the built library with include/magnet_hip.h, not with a literal that goes stale on an ABI bump
    (round 4: the literal said 301 while the header said 302 — the driver's build check would have failed)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "__graft_entry__.py")).read()
    assert "MAGNET_HIP_VERSION" in src and not re.search(r"magnet_version\(\)\s*==\s*\d", src)
