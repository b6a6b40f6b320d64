"""The store-data hazard of VMEM stores wider than 64 bits, checked in the DISASSEMBLY of what was built (replaces round 4's source-text check, which
could not see a new asm statement with another mnemonic). tools/check_store_hazard.py explains the hazard; here: every gfx950 code object of the library is clean, and the
scanner does flag a kernel that is built without the wait states (and accepts the same kernel with them) — so a green run means something."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import check_store_hazard as scanner  # noqa: E402

BROKEN = r'''
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// the asm store of bepu_cluster_kernel.h's store_agent_f4 %s its wait states, followed by arithmetic the compiler is free to place in the store's data registers
__global__ void probe(float4* out, const float* in) {
    f4 x = {in[threadIdx.x], in[threadIdx.x + 64], in[threadIdx.x + 128], in[threadIdx.x + 192]};
    asm volatile("global_store_dwordx4 %%0, %%1, off sc1%s" ::"v"(out + threadIdx.x), "v"(x) : "memory");
    asm volatile("v_add_f32 %%0, %%0, %%0" : "+v"(x.x));
    asm volatile("v_add_f32 %%0, %%0, %%0" : "+v"(x.y));
    out[threadIdx.x + 64] = make_float4(x.x, x.y, x.z, x.w);
}
'''


def build_probe(tmp_path, with_nop: bool) -> str:
    src = tmp_path / ("probe_ok.hip" if with_nop else "probe_broken.hip")
    src.write_text(BROKEN % (("with", "\\n\\ts_nop 1") if with_nop else ("WITHOUT", "")))
    obj = str(src)[:-4] + ".o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", "-o", obj, str(src)], stderr=subprocess.DEVNULL)
    return obj


def test_the_scanner_flags_a_wide_store_without_wait_states_and_accepts_it_with_them(tmp_path):
    hazards, stores = scanner.scan_text(scanner.disassemble(build_probe(tmp_path, with_nop=False)))
    assert stores >= 1 and hazards, "a 128-bit asm store followed at once by a VALU write of its data register must be reported"
    assert "v_add_f32" in hazards[0][2] and hazards[0][3] < 2
    hazards, stores = scanner.scan_text(scanner.disassemble(build_probe(tmp_path, with_nop=True)))
    assert stores >= 1 and not hazards, hazards


def test_every_built_code_object_is_free_of_the_store_data_hazard():
    build_dir = os.path.join(REPO, "bepuphysics2_amd", "csrc", "build")
    objects = sorted(os.path.join(build_dir, f) for f in os.listdir(build_dir) if f.endswith(".o")) if os.path.isdir(build_dir) else []
    if not objects:
        pytest.skip("no object files: the library was not built in this tree (__graft_entry__.build())")
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(lambda o: (o, scanner.scan_text(scanner.disassemble(o))), objects))
    total = 0
    for obj, (hazards, stores) in results:
        assert not hazards, (os.path.basename(obj), hazards[:3])
        total += stores
    shared = [stores for obj, (_, stores) in results if os.path.basename(obj).startswith("bepu_cluster_") and "s" in os.path.basename(obj).split("_")[-1]]
    assert total > 1000 and shared and min(shared) > 50, "the split-plan units publish records with 128-bit stores: the scan must have seen them"
