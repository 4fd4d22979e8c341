#!/usr/bin/env python3
"""VGPRs / spills / scratch / occupancy of the transform kernels of one static plan.
usage: tools/resource_report.py <plan index in SM_STATIC_PLANS> [extra hipcc flags]"""
import re
import subprocess
import sys
from pathlib import Path

csrc = Path(__file__).resolve().parents[1] / "shardmerge_amd" / "csrc"
idx = sys.argv[1] if len(sys.argv) > 1 else "6"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", *sys.argv[2:],
       "-Rpass-analysis=kernel-resource-usage", f"-DSM_PLAN_INDEX={idx}", "-c", "smhip_inst.hip", "-o", f"/tmp/resource_report_{idx}.o"]
txt = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr


def field(blk, key):
    m = re.search(re.escape(key) + r":\s*(\d+)", blk)
    return m.group(1) if m else "?"


for blk in txt.split("Function Name:")[1:]:
    m = re.search(r"sm_kernelINS_(\w+?)INS_5SPlanILi(\d+)ELi(\d+)", blk)
    name = f"{m.group(1)}<{m.group(2)},{m.group(3)}>" if m else blk[:40]
    print(f"{name:22s} VGPRs {field(blk, 'VGPRs'):>4s}  spill {field(blk, 'VGPRs Spill'):>3s}  scratch {field(blk, 'ScratchSize [bytes/lane]'):>4s}"
          f"  waves/SIMD {field(blk, 'Occupancy [waves/SIMD]'):>2s}  LDS {field(blk, 'LDS Size [bytes/block]')}")
