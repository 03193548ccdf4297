"""Static instruction-class counts of the conv kernels' code objects (development aid).
usage: python tools/isa_classes.py [conv4.hip|conv2.hip|frontend.hip] [substring of the template arguments ...]
Compiles the unit to assembly (hipcc -S, the product flags) and prints, per kernel instantiation: matrix, vector, LDS, vector-memory,
scalar-ALU, s_nop, s_waitcnt, branch and barrier instructions in the whole function, and the same for its hottest loop nest
(the innermost loops that contain matrix instructions).  rocprofv3's SQ_INSTS_SALU counts the scalar classes together."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = sys.argv[1] if len(sys.argv) > 1 else "conv4.hip"
pats = sys.argv[2:]
asm = f"/tmp/{unit}.s"
flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-value"] + (["-fno-slp-vectorize"] if unit in ("conv4.hip", "frontend.hip") else [])
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", os.path.join(ROOT, "softspoken_amd", "csrc", unit), "-o", asm], check=True, stderr=subprocess.DEVNULL)
text = open(asm).read()

def cls(i):
    if i.startswith("v_mfma"): return "mfma"
    if i.startswith("s_nop"): return "s_nop"
    if i.startswith("s_waitcnt"): return "s_waitcnt"
    if i.startswith("s_barrier"): return "s_barrier"
    if i.startswith("s_cbranch") or i.startswith("s_branch"): return "s_branch"
    if i.startswith("s_load") or i.startswith("s_buffer"): return "smem"
    if i.startswith("s_"): return "salu"
    if i.startswith("ds_"): return "lds"
    if i.split("_")[0] in ("global", "buffer", "flat", "scratch"): return "vmem"
    if i.startswith("v_"): return "valu"
    return None

def count(lines):
    c = collections.Counter()
    for l in lines:
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        k = cls(t.split()[0])
        if k: c[k] += 1
    return c

for m in re.finditer(r"^(_ZN2ss\w+):\s*; @.*?\n(.*?)s_endpgm", text, flags=re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if pats and not any(p in name for p in pats):
        continue
    body = m.group(2).splitlines()
    c = count(body)
    if not c["mfma"]:
        continue
    sc = c["salu"] + c["s_nop"] + c["s_waitcnt"] + c["s_branch"] + c["s_barrier"] + c["smem"]
    short = re.sub(r"^void ss::", "", name).split("(")[0]
    print(short)
    print("   whole function:", dict(c), " scalar classes / mfma = %.2f" % (sc / c["mfma"]))
    # blocks that contain MFMAs
    blocks, cur = [], []
    for l in body:
        if re.match(r"^\.LBB", l):
            blocks.append(cur); cur = []
        cur.append(l)
    blocks.append(cur)
    hot = collections.Counter()
    for b in blocks:
        cb = count(b)
        if cb["mfma"] >= 4:
            hot.update(cb)
    if hot["mfma"]:
        sc = hot["salu"] + hot["s_nop"] + hot["s_waitcnt"] + hot["s_branch"] + hot["s_barrier"] + hot["smem"]
        print("   blocks with >= 4 mfma:", dict(hot), " scalar classes / mfma = %.2f, valu / mfma = %.2f, lds / mfma = %.2f" % (sc / hot["mfma"], hot["valu"] / hot["mfma"], hot["lds"] / hot["mfma"]))
