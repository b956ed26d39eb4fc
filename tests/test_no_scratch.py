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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_kernel_uses_scratch_memory(tmp_path):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(srcs) >= 9
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        texts = list(ex.map(lambda s: _listing(s, str(tmp_path)), srcs))
    n_kernels, bad = 0, []
    for src, txt in zip(srcs, texts):
        for blk in txt.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if name.startswith("_ZN7rocprim"):
                continue
            n_kernels += 1
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
            if scratch != 0 or spills != 0:
                bad.append((os.path.basename(src), name, scratch, spills))
    assert n_kernels > 250, n_kernels
    assert not bad, f"kernels with scratch memory / spilled vector registers: {bad}"
