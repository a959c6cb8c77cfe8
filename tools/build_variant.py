#!/usr/bin/env python3
"""A second build of libtgcn.so with extra preprocessor flags on ONE source, for interleaved A/B timing against the
default library (tools/ab_dense.py with TGCN_LIB_PATH):

    python tools/build_variant.py pair dense.hip -DTGCN_NT_PAIR_TILES=1
    TGCN_LIB_PATH=pytextgcn_amd/lib/variants/libtgcn_pair.so python tools/ab_dense.py

The other objects are the default build's (pytextgcn_amd/lib/obj); the variant's object and library live under
pytextgcn_amd/lib/variants/ (git-ignored like every binary, shipped to the GPU box with the snapshot)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import build as B  # noqa: E402

name, source, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
out_dir = os.path.join(B.LIB_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"{os.path.splitext(source)[0]}_{name}.o")
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC,
       "-x", "hip", "-c", os.path.join(B.CSRC, source), "-o", obj] + flags
if source == "dense.hip":
    cmd.append("-Rpass-analysis=kernel-resource-usage")
res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
if res.returncode != 0:
    sys.exit(res.stdout[-4000:])
if source == "dense.hip":
    B.check_asm_ring_kernels(res.stdout)             # the variant must not spill in the asm-ring kernels either
objs = [obj if s == source else os.path.join(B.OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES]
lib = os.path.join(out_dir, f"libtgcn_{name}.so")
res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", lib], stdout=subprocess.PIPE,
                     stderr=subprocess.STDOUT, text=True)
if res.returncode != 0:
    sys.exit(res.stdout[-4000:])
print(lib)
