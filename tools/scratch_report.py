"""Kernels of the PRODUCT build that use scratch (or all with --all): compiles every csrc/*.hip with -Rpass-analysis=kernel-resource-usage
(the flags of softspoken_amd/build.py) and prints name, VGPRs, AGPRs, scratch bytes per lane.  usage: python tools/scratch_report.py [--all] [--dev]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import build as B
show_all, dev = "--all" in sys.argv, "--dev" in sys.argv
for src in B.SOURCES:
    cmd = [B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + (["-DSS_DEVBUILD"] if dev and src in B.DEV_SOURCES else []) + \
          ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    name = vg = ag = None
    for l in out.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m: name = m.group(1)
        m = re.search(r" VGPRs: (\d+)", l)
        if m: vg = int(m.group(1))
        m = re.search(r"AGPRs: (\d+)", l)
        if m: ag = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", l)
        if m and (show_all or int(m.group(1)) > 0):
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{src:14s} {dn[:120]:120s} vgpr {vg:3d} agpr {ag:3d} scratch {m.group(1)}")
