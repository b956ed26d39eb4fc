"""Instruction mix of one kernel out of a hipcc -S listing: python scripts/isa_count.py file.s <substring of the mangled name>.
Counts are static (whole kernel text); the tile loops of the hot kernels are straight-line, so they track per-tile issue."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(_Z\S*%s\S*):[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(pat), txt, re.S | re.M)
if not m:
    sys.exit("kernel not found")
body = m.group(2)
ops = collections.Counter()
for line in body.splitlines():
    line = line.strip()
    if not line or line.startswith((";", ".", "//")) or line.endswith(":"):
        continue
    ops[line.split()[0]] += 1
cls = collections.Counter()
for op, n in ops.items():
    if op.startswith("v_mfma"): c = "mfma"
    elif op.startswith("v_"): c = "valu"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
    elif op.startswith("s_waitcnt"): c = "waitcnt"
    elif op.startswith("s_nop"): c = "nop"
    elif op.startswith("s_"): c = "salu"
    else: c = "other"
    cls[c] += n
print(m.group(1)[:100])
print("whole kernel:", dict(cls))
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
print(", ".join(f"{o} {n}" for o, n in ops.most_common(top)))

# the largest basic block (the straight-line decoder + finish of the tile kernels): what one tile issues
blocks = re.split(r"\n\.LBB\d+_\d+:[^\n]*", body)
big = max(blocks, key=lambda b: b.count("\n"))
bops = collections.Counter()
for line in big.splitlines():
    line = line.strip()
    if not line or line[0] in ";./":
        continue
    bops[line.split()[0]] += 1
bcls = collections.Counter()
for op, n in bops.items():
    c = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_")) else "waitcnt" if op.startswith("s_waitcnt") else
         "nop" if op.startswith("s_nop") else "salu")
    bcls[c] += n
print("largest basic block:", dict(bcls))
print(", ".join(f"{o} {n}" for o, n in bops.most_common(top)))
