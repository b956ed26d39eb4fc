"""Per-kernel register / spill / scratch / LDS numbers out of a hipcc -S listing (amdhsa metadata)."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk)
    vals = {k: int(g(k).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
                                               "private_segment_fixed_size", "group_segment_fixed_size") if g(k)}
    print(name[:90], vals)
