"""prints VGPR / SGPR / spill / LDS / scratch of every kernel in a gfx950 assembly file (hipcc -S output)"""
import re
import sys

txt = open(sys.argv[1]).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print("%-70s vgpr %s spill %s sgpr %s lds %s scratch %s" % (name.group(1)[:70] if name else "?", g("vgpr_count"), g("vgpr_spill_count"),
                                                                 g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
