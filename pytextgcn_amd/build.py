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


# Kernels that may use scratch memory, with the reason.  Everything else in csrc/ must not (the build FAILS): a spill is a
# silent 2-10 x on a kernel that was tuned to its register budget, and nothing else would report it.
SCRATCH_ALLOWED = (
    (re.compile(r"rocprim"), "rocPRIM's own kernels (one-sweep radix sort: per-lane digit arrays in private memory), "
                             "one-off plan / graph construction, not on the per-step path"),
)


def resource_usage(remarks: str):
    """[(kernel, scratch bytes per lane, VGPRs spilled, SGPRs spilled)] from hipcc -Rpass-analysis=kernel-resource-usage."""
    out = []
    for b in re.split(r"remark: Function Name: ", remarks)[1:]:
        name = b.split()[0]
        m = [re.search(pat, b) for pat in (r"ScratchSize \[bytes/lane\]: (\d+)", r"VGPRs Spill: (\d+)", r"SGPRs Spill: (\d+)")]
        if m[0] is None or m[1] is None:
            raise RuntimeError(f"build check: the kernel-resource-usage remark of {name} carries no "
                               "'ScratchSize [bytes/lane]' / 'VGPRs Spill' field (has hipcc changed its wording?)")
        out.append((name, int(m[0].group(1)), int(m[1].group(1)), int(m[2].group(1)) if m[2] else 0))
    return out


def check_no_spills(source: str, remarks: str) -> dict:
    """Every kernel of `source`: no scratch memory, no spilled registers -- unless SCRATCH_ALLOWED names it.  Returns the
    summary that goes into the build report; raises (the build fails) on a kernel that is not allow-listed."""
    usage = resource_usage(remarks)
    bad, allowed, sgpr_only = [], [], 0
    for name, scratch, vspill, sspill in usage:
        if not (scratch or vspill):
            # SGPRs "spilled" with 0 bytes of scratch live in lanes of a spare VGPR (v_writelane / v_readlane): no memory
            # traffic; counted in the report, not a failure
            sgpr_only += 1 if sspill else 0
            continue
        line = f"{name}: scratch {scratch} B/lane, {vspill} VGPRs / {sspill} SGPRs spilled"
        (allowed if any(pat.search(name) for pat, _ in SCRATCH_ALLOWED) else bad).append(line)
    if bad:
        raise RuntimeError(f"build check: kernels of {source} use scratch memory / spill registers (give the kernel a smaller "
                           "register footprint or another occupancy target, or allow-list it in build.SCRATCH_ALLOWED with a "
                           "reason):\n  " + "\n  ".join(bad))
    return {"kernels": len(usage), "with_scratch": 0, "allow_listed_with_scratch": len(allowed),
            "sgprs_parked_in_vgpr_lanes": sgpr_only}


def scan_pipe_isa(disassembly: str):
    """(number of k_gemm_pipe functions, violations) in `llvm-objdump -d --no-show-raw-insn` output of the device code: a
    violation is an instruction that names a destination register of a `global_load_dwordx4` issued earlier in the same
    function and not yet covered by an `s_waitcnt vmcnt(0)`."""
    def regs_of(text):
        out = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", text):
            out.add(int(m.group(1)))
        return out
    name, seen, bad = None, 0, []
    pending = set()
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name, pending = (m.group(1) if "k_gemm_pipe" in m.group(1) else None), set()
            seen += name is not None
            continue
        ins = line.split("//")[0].strip()
        if name is None or not ins or ins.startswith(";"):
            continue
        op = ins.split()[0]
        if op == "global_load_dwordx4":
            dst, rest = ins.split(None, 1)[1].split(",", 1)
            if regs_of(rest) & pending:
                bad.append(f"{name}: {ins}")
            pending |= regs_of(dst)
        elif op == "s_waitcnt" and "vmcnt(0)" in ins:
            pending = set()
        elif regs_of(ins) & pending:
            bad.append(f"{name}: {ins}")
    return seen, bad


def check_pipe_kernel_isa(obj: str) -> str:
    """k_gemm_pipe keeps inline-asm loads in flight across a whole k-loop (dense.hip): NO instruction may touch a
    destination register of such a load between its issue and the `s_waitcnt vmcnt(0)` that covers it -- not a use, not a
    copy the register allocator slipped in.  The spill remarks cannot see a copy; the disassembly can.  The device code is
    taken out of the object (llvm-objdump --offloading) and every `k_gemm_pipe` function is scanned in program order (its
    loop bodies are straight-line between the waits).  Returns a one-line summary; raises on a violation.  When the ROCm
    binutils are not where expected the summary starts with "k_gemm_pipe ISA check skipped": the caller (`build`) then does
    NOT ship the unverified kernel -- it rebuilds dense.hip with -DTGCN_NT_PIPE=0 (the class-width nt product falls back on
    k_gemm_tall, whose loads the compiler tracks) and says so loudly."""
    import glob
    import tempfile
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        return "k_gemm_pipe ISA check skipped: llvm-objdump not found"
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "dense.o")
        shutil.copy(obj, local)
        subprocess.run([objdump, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        code = glob.glob(os.path.join(d, "dense.o.*gfx950*"))
        if not code:
            return "k_gemm_pipe ISA check skipped: no gfx950 bundle extracted"
        res = subprocess.run([objdump, "-d", "--no-show-raw-insn", code[0]], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True)
    seen, bad = scan_pipe_isa(res.stdout)
    if seen == 0:
        raise RuntimeError("build check: no k_gemm_pipe function in the disassembly of dense.o (did the kernel's name change?)")
    if bad:
        raise RuntimeError("build check: a register of an inline-asm load in flight is touched before its wait:\n  "
                           + "\n  ".join(bad[:10]))
    return f"k_gemm_pipe ISA check: {seen} instantiations, no instruction touches a ring register in flight"


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
            if s.endswith(".hip"):
                cmd.append("-Rpass-analysis=kernel-resource-usage")
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((s, obj, tmp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    report = _read_report()
    for s, obj, tmp, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed.append(f"{s}:\n{out}")
            if os.path.exists(tmp):
                os.remove(tmp)
            continue
        try:
            if s.endswith(".hip"):
                report[s] = check_no_spills(s, out)
            if s == "dense.hip":
                check_asm_ring_kernels(out)
                note = check_pipe_kernel_isa(tmp)
                if note.startswith("k_gemm_pipe ISA check skipped"):
                    # the kernel cannot be verified on this box: do not ship it
                    import warnings
                    msg = (f"pytextgcn_amd.build: {note}; dense.hip is rebuilt with -DTGCN_NT_PIPE=0 (the class-width nt "
                           "product runs on k_gemm_tall instead of the unverified k_gemm_pipe)")
                    warnings.warn(msg)
                    print(msg, file=sys.stderr)
                    res = subprocess.run([hipcc] + flags + ["-DTGCN_NT_PIPE=0", "-x", "hip", "-c", os.path.join(CSRC, s), "-o", tmp,
                                          "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE,
                                         stderr=subprocess.STDOUT, text=True)
                    if res.returncode != 0:
                        raise RuntimeError(f"{s} (-DTGCN_NT_PIPE=0):\n{res.stdout}")
                    report[s] = check_no_spills(s, res.stdout)
                    check_asm_ring_kernels(res.stdout)
                    note += "; built with -DTGCN_NT_PIPE=0"
                report[s]["pipe_kernel_isa_check"] = note
                if verbose:
                    print(note, file=sys.stderr)
        except RuntimeError as e:
            failed.append(str(e))
            if os.path.exists(tmp):
                os.remove(tmp)
            continue
        os.replace(tmp, obj)
    if failed:
        raise RuntimeError("hipcc failed building libtgcn.so:\n" + "\n".join(failed))
    n_k = sum(v.get("kernels", 0) for v in report.values())
    n_allowed = sum(v.get("allow_listed_with_scratch", 0) for v in report.values())
    report_line = (f"{n_k} kernels in {len(report)} sources, 0 kernels of csrc/ with scratch"
                   + (f" ({n_allowed} library kernels of rocPRIM allow-listed)" if n_allowed else ""))
    if verbose:
        print("build check: " + report_line, file=sys.stderr)
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
    import json
    with open(REPORT_PATH, "w") as f:
        json.dump({"sources": report, "summary": report_line}, f, indent=1, sort_keys=True)
    return LIB_PATH


REPORT_PATH = os.path.join(LIB_DIR, "build_report.json")


def _read_report() -> dict:
    """Per-source results of the last build's checks (a source that is not recompiled keeps its entry)."""
    import json
    try:
        with open(REPORT_PATH) as f:
            return dict(json.load(f).get("sources", {}))
    except (OSError, ValueError):
        return {}


def build_report() -> dict:
    """What the checks of the build that produced lib/libtgcn.so found: {"summary": "... 0 kernels with scratch ...",
    "sources": {source: {"kernels", "with_scratch", "allow_listed_with_scratch"[, "pipe_kernel_isa_check"]}}}; empty when the
    library was built by something else."""
    import json
    try:
        with open(REPORT_PATH) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
