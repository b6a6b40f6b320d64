"""Per-function register use of one cluster unit, from hipcc's kernel-resource-usage remarks (developer tool).
    python tools/unit_regs.py bepu_cluster_hot_1024.hip [-DFLAG ...]      prints name, VGPRs, scratch bytes, spilled VGPRs; then the .text size of the gfx950 code object"""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "bepuphysics2_amd", "csrc")
unit, extra = sys.argv[1], sys.argv[2:]
obj = tempfile.mktemp(suffix=".o")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Xarch_device", "-fno-slp-vectorize", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
       "-Wno-array-bounds", "-c", "-o", obj, unit, "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
def demangle(n):
    try:
        return subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], text=True).strip()
    except Exception:
        return n
for r in rows:
    name = demangle(r["name"])
    if "cluster" not in name and "Contact" not in name and "run_" not in name and "requirk" not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short)[:150]
    print(f"{r.get('vgprs', 0):4d} vgprs {r.get('scratch', 0):5d} B scratch {r.get('spill', 0):4d} spilled  {short}")
sys.path.insert(0, os.path.join(REPO, "tools"))
import check_store_hazard as c
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(c.LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
    out = subprocess.check_output([os.path.join(c.LLVM, "llvm-readelf"), "-S", "-sW", co], text=True)
    for line in out.splitlines():
        if " .text " in line:
            print("  .text bytes:", int(line.split()[5 if line.split()[0] == '[' else 4], 16) if False else line.split())
    sizes = []
    for line in out.splitlines():
        cols = line.split()
        if len(cols) >= 8 and cols[3] == "FUNC":
            sizes.append((int(cols[2]), cols[7]))
    for n, name in sorted(sizes, reverse=True)[:12]:
        print(f"  {n:8d} bytes  {re.sub(r'[(].*', '', demangle(name))[:140]}")
os.remove(obj)
