"""HBM bytes per launch of the dominant kernel of a command, the way MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
(no tracing alongside), KiB units, FETCH_SIZE doubled on gfx950. Usage: python tools/pmc_traffic.py [--kernel SUBSTRING] -- <command ...>"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

argv = sys.argv[1:]
want = None
if argv and argv[0] == "--kernel":
    want, argv = argv[1], argv[2:]
if argv and argv[0] == "--":
    argv = argv[1:]
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix="bepu_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + argv, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   timeout=600, check=True)
    per_kernel = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and (want is None or want in r["Kernel_Name"]):
                a = per_kernel.setdefault(r["Kernel_Name"], [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    shutil.rmtree(d, ignore_errors=True)
    out[counter] = per_kernel
for k in sorted(out["FETCH_SIZE"], key=lambda k: -out["FETCH_SIZE"][k][0])[:4]:
    f, n = out["FETCH_SIZE"][k]
    w, _ = out["WRITE_SIZE"].get(k, (0.0, 1))
    fetch, write = f / n * 1024.0, w / n * 1024.0
    print(f"{k[:110]}: {n} launches, FETCH_SIZE {fetch / 1e6:.1f} MB raw -> {2 * fetch / 1e6:.1f} MB (x2, gfx950), WRITE_SIZE {write / 1e6:.1f} MB, traffic {(2 * fetch + write) / 1e6:.1f} MB per launch")
