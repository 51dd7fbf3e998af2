#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the built library (llvm-objdump --offloading + llvm-readelf --notes):
   python profiles/kernel_regs.py [lib.so] > table.  Used to check that a refactoring changed no kernel's budget."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(lib):
    d = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(d)
    try:
        import shutil
        shutil.copy(os.path.join(cwd, lib), os.path.join(d, "lib.so"))          # the bundles are extracted next to the input
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], capture_output=True, check=True)
        rows = {}
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                rows[name] = (g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                              g("group_segment_fixed_size"))
        return rows
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else "spamtree_amd/libspamtree_hip.so"
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spill':>5} {'scratch':>7} {'lds':>7}  kernel")
    for name, r in sorted(kernel_table(lib).items()):
        print(f"{r[0]:>5} {r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>7} {r[5]:>7}  {name[:150]}")
