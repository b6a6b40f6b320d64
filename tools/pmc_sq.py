"""What bounds a kernel when it is not bytes: the SQ counters of the dominant kernel of a command (GPU box). Two rocprofv3 --pmc passes (8 SQ slots each, no tracing
alongside), averaged per launch, and the fractions MI355X_MICROARCH.md's PMC section defines: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles),
VALU busy = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x GRBM_GUI_ACTIVE / 8 XCDs). Usage: python tools/pmc_sq.py [--kernel SUBSTRING] -- <command ...>"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

argv = sys.argv[1:]
want = "cluster_kernel"
if argv and argv[0] == "--kernel":
    want, argv = argv[1], argv[2:]
if argv and argv[0] == "--":
    argv = argv[1:]
SETS = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE",
        "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_SALU"]
values = {}
launches = 0
for counters in SETS:
    d = tempfile.mkdtemp(prefix="bepu_sq_", dir="/tmp")
    subprocess.run(["rocprofv3", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", "pmc", "--"] + argv, env=dict(os.environ, TMPDIR="/tmp"),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=False)
    total, count = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"] and int(r.get("Grid_Size", "0") or 0) >= 64 * 200:  # the step launches, not a tail or warm-up grid of a few workgroups
                total[r["Counter_Name"]] = total.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                count[r["Counter_Name"]] = count.get(r["Counter_Name"], 0) + 1
    shutil.rmtree(d, ignore_errors=True)
    for k in total:
        values[k] = total[k] / count[k]
        launches = max(launches, count[k])
if not values:
    print("no counters collected")
    sys.exit(1)
print(f"{want}: {launches} launches; per launch:")
for k in sorted(values):
    print(f"  {k:24s} {values[k]:16.0f}")
cus = int(os.environ.get("CUS", "256"))
gui = values.get("GRBM_GUI_ACTIVE", 0.0)
wave = values.get("SQ_WAVE_CYCLES", 0.0)
if gui and wave:
    xcds = int(os.environ.get("XCDS", "8"))
    simd_quads = gui / xcds / 4.0 * cus * 4  # quad-cycles all SIMDs of the chip offer during the launch (GRBM_GUI_ACTIVE is summed over the XCDs: value / 8 / launch time = the shader clock)
    print(f"  waves resident per SIMD (SQ_WAVE_CYCLES / SIMD quad-cycles)      {wave / simd_quads:.2f}")
    for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if name in values:
            print(f"  {name} / SQ_WAVE_CYCLES {values[name] / wave:6.3f}    per SIMD: {values[name] / simd_quads:6.3f} of its cycles")
    if "SQ_INSTS_VALU" in values:
        print(f"  VALU instructions per launch {values['SQ_INSTS_VALU']:.0f}; x 4 cycles / (SIMDs x cycles) = {values['SQ_INSTS_VALU'] * 4 / (gui / xcds * cus * 4):.3f} of the chip's VALU issue slots ({cus} CUs)")
