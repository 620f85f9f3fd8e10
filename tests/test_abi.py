"""C-ABI checks that need no GPU: the shared library loads, exports every entry point include/storygen_hip.h declares,
the ctypes table in storygen_amd/_lib.py covers exactly those names with matching arity, the descriptor structs have
the C layout, and host-side validation rejects bad descriptors before anything would be enqueued."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "storygen_hip.h")


@pytest.fixture(scope="module")
def lib():
    from storygen_amd import _lib
    from storygen_amd.build import build
    build(force=False, verbose=False)      # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def _header_text():
    src = open(HEADER).read()
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def _declared_functions():
    """name -> number of parameters, parsed from the header."""
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const\s+char\s*\*)\s+(sg_\w+)\s*\(([^;{]*?)\)\s*;", _header_text(), flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_header_declares_the_expected_surface():
    fns = _declared_functions()
    for need in ("sg_gemm_f16", "sg_conv3x3_nhwc_f16", "sg_attn_fwd_f16", "sg_groupnorm_nhwc_f16", "sg_layernorm_f16",
                 "sg_timestep_embed_f32", "sg_cfg_ddim_step_f32", "sg_version", "sg_last_error", "sg_device_arch"):
        assert need in fns, need
    assert len(fns) >= 20


def test_library_exports_every_declared_symbol(lib):
    from storygen_amd._lib import LIB_PATH
    nm = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    missing = sorted(set(_declared_functions()) - exported)
    assert not missing, f"declared in the header but not exported: {missing}"
    stray = sorted(s for s in exported if s.startswith("sg_") and s not in _declared_functions())
    assert not stray, f"exported sg_* symbols missing from the header: {stray}"


def test_ctypes_table_matches_header(lib):
    from storygen_amd._lib import SIGNATURES
    fns = _declared_functions()
    assert set(SIGNATURES) == set(fns)
    for name, (_, args) in SIGNATURES.items():
        assert len(args) == fns[name], (name, len(args), fns[name])
        assert getattr(lib, name) is not None


def test_descriptor_layouts_match_a_c_compiler(tmp_path):
    """sizeof / offsetof of every descriptor as gcc sees the header vs the ctypes Structures."""
    from storygen_amd import _lib
    structs = {"sg_gemm_desc": _lib.GemmDesc, "sg_conv3x3_desc": _lib.ConvDesc, "sg_attn_desc": _lib.AttnDesc,
               "sg_groupnorm_desc": _lib.GroupNormDesc, "sg_groupnorm_bwd_desc": _lib.GroupNormBwdDesc,
               "sg_attn_bwd_desc": _lib.AttnBwdDesc, "sg_adamw_desc": _lib.AdamWDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in st._fields_:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)], check=True)   # header is plain C
    got = dict(ln.split() for ln in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == C.sizeof(st), cname
        for f, _ in st._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(st, f).offset, (cname, f)


def test_version_and_error_string(lib):
    assert lib.sg_version() == 1
    assert isinstance(lib.sg_last_error(), bytes)


def test_multiply_shift_divisors_are_exact(lib):
    """The GEMM / conv prologues divide by tile counts, image sizes and channel chunks with host-made multiply-shift constants
    (gemm_conv.hip make_fastdiv / fd_div, the same code on host and device): exact for every 32-bit numerator."""
    assert lib.sg_debug_fastdiv_selftest() == 0


def test_host_validation_rejects_before_launch(lib):
    """Every entry point validates on the host first: these calls return SG_EINVAL / SG_EUNSUP without touching a
    device (so they work on a GPU-less box) and leave a message in sg_last_error()."""
    from storygen_amd._lib import AttnDesc, ConvDesc, GemmDesc, GroupNormDesc
    assert lib.sg_gemm_f16(None, None) == -1
    assert b"null" in lib.sg_last_error()
    d = GemmDesc()
    d.A, d.W, d.C = 0x1000, 0x2000, 0x3000
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = 64, 64, 60, 64, 64, 64          # K not a multiple of 8
    assert lib.sg_gemm_f16(C.byref(d), None) == -1
    assert b"multiples of 8" in lib.sg_last_error()
    d.K = 64
    d.A = 0x1004                                                          # misaligned
    assert lib.sg_gemm_f16(C.byref(d), None) == -1
    d.A, d.epilogue = 0x1000, 7
    assert lib.sg_gemm_f16(C.byref(d), None) == -1
    c = ConvDesc()
    c.x, c.w, c.y = 0x1000, 0x2000, 0x3000
    c.B, c.H, c.W, c.Cin, c.Cout, c.ldx, c.ldy, c.stride = 1, 8, 8, 60, 64, 64, 64, 1
    assert lib.sg_conv3x3_nhwc_f16(C.byref(c), None) == -1               # Cin % 64
    c.Cin, c.stride = 64, 3
    assert lib.sg_conv3x3_nhwc_f16(C.byref(c), None) == -1
    a = AttnDesc()
    a.q = a.k = a.vt = a.o = 0x1000
    a.B, a.H, a.Nq, a.Nk, a.D = 1, 8, 64, 64, 64
    a.ldq = a.ldk = a.ldvt = a.ldo = 512
    assert lib.sg_attn_fwd_f16(C.byref(a), None) == -2                   # SG_EUNSUP: head dim
    assert b"head dim" in lib.sg_last_error()
    g = GroupNormDesc()
    assert lib.sg_groupnorm_nhwc_f16(C.byref(g), None) == -1
    assert lib.sg_gemm_workspace_bytes(128, 256, 4) == 128 * 256 * 4 * 4
    assert lib.sg_gemm_workspace_bytes(128, 256, 1) == 0


def test_backward_entry_points_validate_on_the_host(lib):
    """Same for the backward-pass entry points (BASELINE config 4): bad descriptors are refused before any launch."""
    from storygen_amd._lib import AttnBwdDesc, AttnDesc, GroupNormBwdDesc
    P = 0x10000
    assert lib.sg_layernorm_bwd_f16(P, 320, 1, P, 320, P, None, 0, None, 0, None, 0, 1.0, P, 320, 16, 324, 1e-5, None) == -1   # C % 8
    assert lib.sg_layernorm_bwd_f16(P, 320, 1, P, 320, P, P, 320, None, 0, None, 0, 1.0, P, 320, 16, 320, 1e-5, None) == -1  # dy2 w/o gamma2
    assert b"go together" in lib.sg_last_error()
    assert lib.sg_geglu_bwd_f16(P, 96, P, 48, P, 96, 8, 96, None) == -1                    # N8 % 64
    assert lib.sg_transpose_f16(P, 320, 0, P, 12, 12, 320, None) == -1                     # M % 8
    assert lib.sg_transpose_batched_f16(P, 320, 320 * 16, 0, P, 16, 320 * 16, 0, 16, 320, None) == -1      # B = 0
    assert lib.sg_transpose_batched_f16(P, 320, 320 * 16 + 4, 0, P, 16, 320 * 16, 2, 16, 320, None) == -1  # batch stride % 8
    assert lib.sg_sum2x2_f32(P, 6, P, 6, 1, 4, 4, 6, 0, None) == -1                        # C % 4
    assert lib.sg_zero_stuff_f16(P, 320, 0, P, 320, 1, 4, 4, 324, None) == -1
    assert lib.sg_mse_grad_f32(P, P, P, P, P, 0, None) == -1
    assert lib.sg_attn_fwd_lse_f16(C.byref(AttnDesc()), None, None) == -1 and b"lse2" in lib.sg_last_error()
    g = GroupNormBwdDesc()
    g.x = g.dy = g.gamma = g.beta = g.workspace = g.out = P
    g.B, g.HW, g.C, g.groups, g.ldx, g.lddy, g.ldo, g.out_f32 = 1, 64, 32, 32, 32, 32, 32, 1
    assert lib.sg_groupnorm_bwd_nhwc_f16(C.byref(g), None) == -2 and b"channels per group" in lib.sg_last_error()   # cpg = 1
    g.C, g.ldx, g.lddy, g.ldo, g.workspace_bytes = 320, 320, 320, 320, 0
    assert lib.sg_groupnorm_bwd_nhwc_f16(C.byref(g), None) == -1 and b"workspace" in lib.sg_last_error()
    assert lib.sg_groupnorm_bwd_workspace_bytes(4, 32) == 2 * lib.sg_groupnorm_workspace_bytes(4, 32)
    a = AttnBwdDesc()
    a.q = a.k = a.v = a.dout = a.ld2 = a.kt = a.dq = a.qt = a.dot = a.dkt = a.dvt = P
    a.B, a.H, a.Nq, a.Nk, a.D = 1, 8, 60, 77, 40
    for f in ("ldq", "ldk", "ldv", "lddo", "lddq"):
        setattr(a, f, 320)
    a.ldkt = 72                                                       # dq accepts any Nk, but kt rows must cover Nk rounded up to 8
    assert lib.sg_attn_bwd_dq_f16(C.byref(a), None) == -1 and b"ldkt" in lib.sg_last_error()
    a.ldqt = a.lddot = 64
    a.lddkt = a.lddvt = 80
    assert lib.sg_attn_bwd_dkv_f16(C.byref(a), None) == -1 and b"multiple of 8" in lib.sg_last_error()   # Nq = 60
    a.D = 64
    assert lib.sg_attn_bwd_dq_f16(C.byref(a), None) == -2


def test_product_path_has_no_cpu_fallback():
    """The package's compute modules never import the oracle, and ops refuse CPU tensors."""
    import torch
    pkg = os.path.join(ROOT, "storygen_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn
    from storygen_amd import ops
    x = torch.zeros(8, 64, dtype=torch.float16)
    with pytest.raises(TypeError):
        ops.gemm(x, x, torch.zeros(8, 8, dtype=torch.float16))


def test_no_kernel_uses_scratch_memory():
    """The build records every kernel's register / LDS / scratch footprint (hipcc -Rpass-analysis=kernel-resource-usage,
    storygen_amd/lib/kernel_resources.json).  A private segment means spills or an in-memory array — a 6x cliff when it
    happened to the D=160 attention instantiation — so no kernel may have one, and wave64 x 1024-thread launches must fit."""
    import json
    from storygen_amd import build as B
    B.build(verbose=False)
    with open(B.RESOURCES) as f:
        res = json.load(f)
    assert len(res) >= 40
    assert {v["source"] for v in res.values()} == set(B.SOURCES)
    # one listed exception (round 6): mma_fat_kernel, the eight-waves-of-128x64 prototype that is reachable by tile hint only (a measured
    # negative): 128 accumulator registers of its 256 leave the EPILOGUE ~60 dwords short; its mainloop has no spill
    # (storygen_amd/build.py holds the same allowance)
    bad = {k: v for k, v in res.items() if v["scratch_bytes_per_lane"] != 0 and not ("mma_fat_kernel" in k and v["scratch_bytes_per_lane"] <= 256)}
    assert not bad, bad
    assert sum("mma_fat_kernel" in k for k in res) == 4
    assert all(v["vgprs"] <= 256 and v["agprs"] <= 256 and v["lds_bytes_per_block"] <= 160 * 1024 for v in res.values())


def test_library_never_reads_the_environment():
    """Development options live behind sg_debug_set_option (kernel selection must not depend on environment
    variables): no getenv in any kernel source, unknown option names are rejected, "reset" restores the defaults."""
    import glob
    import os
    from storygen_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "storygen_amd", "csrc", "*")):
        with open(path) as f:
            assert "getenv" not in f.read(), path
    lib = _lib.load()
    assert lib.sg_debug_set_option(b"no_split", 1) == 0 and lib.sg_debug_set_option(b"reset", 0) == 0
    assert lib.sg_debug_set_option(b"no_such_option", 1) != 0


def test_shipped_kernels_contain_no_packed_fp32_instructions(tmp_path):
    """one half of a `v_pk_add_f32 ... op_sel` lost a term under the two-branch graph (the only run-to-run difference this
    project ever saw); the library is built with -fno-slp-vectorize.  Checked on the ISA that ships: every gfx950 code object of the
    shared library is disassembled — none holds a packed fp32 add / mul / fma, and the MFMA kernels are really in there."""
    import shutil
    from storygen_amd import build as B
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    B.build(verbose=False)
    lib = tmp_path / "lib.so"
    shutil.copy(B.LIB, lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, capture_output=True, check=True)
    objs = sorted(p for p in tmp_path.iterdir() if "gfx950" in p.name)
    assert len(objs) == len(B.SOURCES)
    packed = mfma = 0
    for o in objs:
        asm = subprocess.run([objdump, "-d", str(o)], capture_output=True, text=True, check=True).stdout
        packed += len(re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", asm))
        mfma += len(re.findall(r"\bv_mfma_", asm))
    assert packed == 0, f"{packed} packed fp32 instructions in the shipped kernels"
    assert mfma > 4000
