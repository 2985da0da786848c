"""Register report of a device-only assembly listing (hipcc --cuda-device-only -S): python scripts/regs.py file.s [filter]
-> kernel name, VGPRs, AGPRs, spilled VGPRs, LDS bytes (from the amdhsa.kernels metadata)."""
import re
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for block in text.split("  - .agpr_count:")[1:]:
    get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, block)
    name = get("name").group(1)
    if flt and flt not in name:
        continue
    agpr = block.split("\n", 1)[0].strip()
    print("%-90s vgpr %3s agpr %3s spill %3s lds %6s" % (name[:90], get("vgpr_count").group(1), agpr, get("vgpr_spill_count").group(1),
                                                          get("group_segment_fixed_size").group(1)))
