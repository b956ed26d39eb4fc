"""No kernel of libpinhip may use scratch memory (spilled registers or dynamically indexed private arrays): every translation
unit is compiled to gfx950 assembly (hipcc -S, device side only, no GPU needed) and the amdhsa metadata of every kernel is
read.  rocPRIM's radix-sort kernels, which maint.hip instantiates, are library code and are exempt.
(`profiles/r04_kernel_resources.txt` is the same listing in full: registers, spills, LDS per kernel.)"""
import glob
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pin_slam_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _listing(src, out_dir):
    out = os.path.join(out_dir, os.path.basename(src)[:-4] + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-S", "--cuda-device-only",
                    src, "-o", out], check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    return open(out).read()


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{mangled kernel name: (file, vgprs, spilled vgprs, scratch bytes, LDS bytes)} of every kernel of the library."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(srcs) >= 9
    out_dir = str(tmp_path_factory.mktemp("asm"))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        texts = list(ex.map(lambda s: _listing(s, out_dir), srcs))
    table = {"__texts__": dict(zip((os.path.basename(x) for x in srcs), texts))}
    for src, txt in zip(srcs, texts):
        for blk in txt.split("  - .agpr_count:")[1:]:
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            table[name] = (os.path.basename(src), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                           g("group_segment_fixed_size"))
    return table


def test_no_kernel_uses_scratch_memory(kernels):
    ours = {k: v for k, v in kernels.items() if not k.startswith("_ZN7rocprim") and k != "__texts__"}
    assert len(ours) > 250, len(ours)
    bad = [(v[0], k, v[3], v[2]) for k, v in ours.items() if v[3] != 0 or v[2] != 0]
    assert not bad, f"kernels with scratch memory / spilled vector registers: {bad}"


def test_register_budgets_of_the_hot_kernels(kernels):
    """The occupancy each hot kernel was tuned for, as a register count (512 vector registers per SIMD lane: <= 256 -> 2 waves
    per SIMD, <= 168 -> 3, <= 72 -> 7).  Guards against silent inflation -- e.g. a second `extern __shared__` symbol in
    sdf.hip took the registration tile kernel from 187 to 211 registers and added 30 address instructions."""
    def one(prefix):
        hit = [(k, v) for k, v in kernels.items() if k.startswith(prefix) and k != "__texts__"]
        assert len(hit) == 1, (prefix, [k for k, _ in hit])
        return hit[0][1]
    # Tracker.tracking at 4 x 64 (the kernel `roofline` is stated on): two waves per SIMD.  (r06: 187 -> 243 with the pivoted
    # rare path of the gather, gn_quad.h quad_gather_pass -- the scheduler spends what the occupancy target leaves; same launch
    # time, 37.5 us, profiles/r06_bench.json -- so the bound is the occupancy step itself.)
    assert one("_ZN3pin25gn_accumulate_quad_kernelILi64ELb0ELb1ELi4ELb0ELi512EEE")[1] <= 256
    # its search: seven waves per SIMD
    assert one("_ZN3pin23knn_brick_listed_kernelILi11ELi0EEE")[1] <= 72
    # Mapper.mapping's tile kernel: three waves per SIMD
    assert one("_ZN3pin18train_fused_kernelILi64ELi4ELi1EEE")[1] <= 168
    # forward-only queries (Mesher.query_points): three waves per SIMD
    assert one("_ZN3pin21sdf_query_quad_kernelILi64ELb0ELi4ELb0ELi1EEE")[1] <= 168
    # the C5 colour registration at 1 x 64: two waves per SIMD
    assert one("_ZN3pin25gn_accumulate_quad_kernelILi64ELb0ELb1ELi1ELb1ELi512EEE")[1] <= 256


def test_the_benchmarked_kernel_runs_on_the_matrix_cores(kernels):
    """gn_accumulate_quad_kernel<64, false, true, 4>: the split-fp16 decoder sweeps are MFMA instructions written out by hand
    (mlp_h2.h), not a library call -- 150 v_mfma_f32_16x16x32_f16 (hidden layers, forward and transposed: 3 products per
    split value) and 12 v_mfma_f32_16x16x16_f16 (the 11-wide input layer and its transpose)."""
    txt = kernels["__texts__"]["sdf.hip"]
    name = "_ZN3pin25gn_accumulate_quad_kernelILi64ELb0ELb1ELi4ELb0ELi512EEE"
    start = txt.index("\n" + name)
    body = txt[start:txt.index("s_endpgm", start)]
    assert body.count("v_mfma_f32_16x16x32_f16") >= 150
    assert body.count("v_mfma_f32_16x16x16_f16") >= 12
    assert "scratch_" not in body
