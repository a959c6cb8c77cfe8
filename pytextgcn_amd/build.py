"""Builds libtgcn.so (the HIP kernels + C ABI of include/tgcn.h) in-tree for gfx950.

`hipcc` cross-compiles without a GPU, so this runs in the build container; the resulting
pytextgcn_amd/lib/libtgcn.so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtgcn.so")
SOURCES = ["spmm.hip", "colsum.hip", "rows.hip", "train.hip", "dense.hip", "plan.hip", "graphbuilder.hip", "error.cpp"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(_ROOT, "include", "tgcn.h")]


OBJ_DIR = os.path.join(LIB_DIR, "obj")


def _newer(path: str, deps) -> bool:
    """True when `path` is missing or older than one of `deps`."""
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _stale() -> bool:
    return _newer(LIB_PATH, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


# Kernels that feed their operand through an INLINE-ASM load ring with hand-counted s_waitcnt (dense.hip: the fully
# unrolled k_gemm_tall<.., NQ > 0, ..>, k_gemm_tall_split and the block-pipelined k_gemm_pipe): the compiler's wait-count pass does not know those
# registers are pending, so a spill or a scratch copy of a ring register between issue and wait would read stale data
# without any diagnostic.  The build therefore FAILS when one of them uses scratch memory or spills vector registers
# (hipcc -Rpass-analysis=kernel-resource-usage).
_RING_KERNEL = re.compile(r"(k_gemm_tall_split|k_gemm_pipe|11k_gemm_tallILi\d+ELb[01]ELb[01]ELi[1-9]\d*E)")


def check_asm_ring_kernels(remarks: str) -> None:
    blocks = re.split(r"remark: Function Name: ", remarks)[1:]
    seen, bad = 0, []
    for b in blocks:
        name = b.split()[0]
        if not _RING_KERNEL.search(name):
            continue
        seen += 1
        m_scratch = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
        m_vspill = re.search(r"VGPRs Spill: (\d+)", b)
        if m_scratch is None or m_vspill is None:
            # another hipcc may word its remarks differently: say so instead of dying on `None.group`
            raise RuntimeError(f"build check: the kernel-resource-usage remark of {name} carries no "
                               "'ScratchSize [bytes/lane]' / 'VGPRs Spill' field (has hipcc changed its wording?)")
        scratch, vspill = int(m_scratch.group(1)), int(m_vspill.group(1))
        if scratch or vspill:
            bad.append(f"{name}: scratch {scratch} B/lane, {vspill} VGPRs spilled")
    if seen == 0:
        raise RuntimeError("build check: no kernel-resource-usage remarks for the asm-ring kernels of dense.hip "
                           "(did the kernel names change?)")
    if bad:
        raise RuntimeError("build check: kernels with an inline-asm load ring must not spill (stale ring registers):\n  "
                           + "\n  ".join(bad))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is newer than its object file (one hipcc process per stale source, run side by side),
    then link.  Returns the library path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libtgcn.so cannot be built (ROCm toolchain required)")
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
             "-I" + os.path.join(_ROOT, "include"), "-I" + CSRC]
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + HEADERS):
            tmp = obj + ".tmp%d" % os.getpid()
            cmd = [hipcc] + flags + ["-x", "hip", "-c", src, "-o", tmp]
            if s == "dense.hip":
                cmd.append("-Rpass-analysis=kernel-resource-usage")
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((s, obj, tmp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for s, obj, tmp, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed.append(f"{s}:\n{out}")
            if os.path.exists(tmp):
                os.remove(tmp)
        else:
            if s == "dense.hip":
                try:
                    check_asm_ring_kernels(out)
                except RuntimeError as e:
                    failed.append(str(e))
                    os.remove(tmp)
                    continue
            os.replace(tmp, obj)
    if failed:
        raise RuntimeError("hipcc failed building libtgcn.so:\n" + "\n".join(failed))
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed linking libtgcn.so:\n" + res.stdout)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
