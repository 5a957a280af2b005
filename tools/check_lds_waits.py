#!/usr/bin/env python3
"""Build-time checks on librtsync.so's device code: (1) no instruction reads the destination of an LDS load before an
s_waitcnt that retires it; (2) no DPP instruction reads a VGPR within two wait states of the VALU write that produced it
(check_dpp below).

Why: csrc/otw.hip issues some LDS reads through inline asm (immediate-offset ds_read_b64 in strip_chain); the compiler's
own waitcnt insertion does not see those, the code waits for them explicitly (chain_wait) behind a 16-way switch.  That is
only correct as long as the register allocator places no copy of a loaded register inside a case, i.e. in front of the
wait -- a property of the generated code, so it is checked on the generated code.

Method: disassemble the gfx950 code object, and per kernel scan linearly: every ds_read* puts its destination registers
into a FIFO (LDS operations return in order: `s_waitcnt lgkmcnt(n)` retires all but the n most recent ones; scalar
memory loads, which share the counter and return out of order, conservatively make only lgkmcnt(0) retire anything while
one is outstanding); any instruction that names a pending destination register as an operand before it is retired is a
violation.  Branches do not reset the scan (the code falls through case by case in address order, and a wait that only
exists behind a join is exactly what must not be relied upon by an earlier use), except that an unconditional s_branch /
s_endpgm ends a straight-line region: what follows is reached from elsewhere, so the FIFO restarts empty there.

    python tools/check_lds_waits.py [path/to/librtsync.so] [kernel-name-substring]     exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_WAIT = re.compile(r"lgkmcnt\((\d+)\)")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(so_path):
    """Text of every gfx950 code object bundled in the shared library (one per translation unit)."""
    import glob
    import shutil
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(so_path, os.path.join(d, "lib.so"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = sorted(glob.glob(os.path.join(d, "lib.so.*gfx950*")))
        if not cos:
            raise RuntimeError("no gfx950 code object found in %s" % so_path)
        return "\n".join(subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
                         for co in cos)


def check(asm_text, only=None):
    """-> (kernels scanned, LDS loads seen, list of violations)."""
    kernels = loads = 0
    bad = []
    name = None
    fifo = []        # pending LDS loads, oldest first: sets of destination VGPRs
    smem = 0         # outstanding scalar loads (share lgkmcnt, return out of order)
    for line in asm_text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name = m.group(1)
            fifo, smem = [], 0
            if only is None or only in name:
                kernels += 1
            continue
        if name is None or (only is not None and only not in name):
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        op, _, rest = ins.partition(" ")
        if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute"):
            dst, _, srcs = rest.partition(",")
            pending = set().union(*fifo) if fifo else set()
            used = _regs(srcs) & pending
            if used:
                bad.append((name, ins, sorted(used)))
            fifo.append(_regs(dst))
            loads += 1
            continue
        if op.startswith("ds_"):          # LDS stores / atomics without a result: they take a counter slot too
            pending = set().union(*fifo) if fifo else set()
            used = _regs(rest) & pending
            if used:
                bad.append((name, ins, sorted(used)))
            fifo.append(set())
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            smem += 1
            continue
        if op == "s_waitcnt":
            w = _WAIT.search(rest)
            if w is not None or "lgkmcnt" not in rest and re.fullmatch(r"\s*(0x[0-9a-f]+|\d+)\s*", rest or ""):
                n = int(w.group(1)) if w else 0
                if smem and n > 0:
                    continue              # out-of-order scalar loads outstanding: only a full wait proves anything
                if n == 0:
                    fifo, smem = [], 0
                elif n < len(fifo):
                    fifo = fifo[len(fifo) - n:]
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            fifo, smem = [], 0
            continue
        if fifo:
            pending = set().union(*fifo)
            used = _regs(rest) & pending
            # a VALU / VMEM instruction that overwrites a pending register without reading it is equally wrong
            if used:
                bad.append((name, ins, sorted(used)))
    return kernels, loads, bad


_DPP_CTRL = (" wave_sh", " wave_ro", " row_sh", " row_ro", " row_bcast", " row_mirror", " row_half_mirror", " quad_perm", " row_newbcast")


def check_dpp(asm_text):
    """Second property of the generated code: a DPP instruction reads its source VGPR no sooner than two wait states
    after a VALU instruction wrote it (gfx940/gfx950 manual hazard).  The compiler places the s_nop itself -- also after
    inline asm that defines the register (csrc's v_min_f64) -- but not *inside* inline asm, and a hand-written DPP move
    would need its own; this scan keeps both honest.  -> (DPP instructions seen, violations)."""
    seen = 0
    bad = []
    name = None
    hist = []
    for line in asm_text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name, hist = m.group(1), []
            continue
        ins = line.split("//")[0].strip()
        if not ins or name is None:
            continue
        op, _, rest = ins.partition(" ")
        if op.startswith("v_") and any(c in rest for c in _DPP_CTRL):
            seen += 1
            srcs = rest.split(",", 1)[1] if "," in rest else ""
            for c in _DPP_CTRL:
                srcs = srcs.split(c)[0]
            src = _regs(srcs)
            ws = 0
            for pop, prest in reversed(hist):
                if ws >= 2:
                    break
                if pop == "s_nop":
                    ws += int(prest.strip() or "0", 0) + 1
                    continue
                if pop.startswith("v_") and _regs(prest.split(",")[0]) & src:
                    bad.append((name, "%s %s" % (pop, prest), ins))
                ws += 1
        hist.append((op, rest))
        hist = hist[-4:]
    return seen, bad


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "real_time_audio_sync_amd", "librtsync.so")
    only = sys.argv[2] if len(sys.argv) > 2 else None
    text = disassemble(so)
    kernels, loads, bad = check(text, only)
    print("%d kernels, %d LDS loads scanned, %d violations" % (kernels, loads, len(bad)))
    for name, ins, regs in bad[:40]:
        print("  %s\n      %s   <- uses v%s before its LDS data is waited for" % (name[:100], ins, regs))
    n_dpp, bad_dpp = check_dpp(text)
    print("%d DPP instructions scanned, %d read a VGPR within two wait states of its VALU write" % (n_dpp, len(bad_dpp)))
    for name, w, r in bad_dpp[:40]:
        print("  %s\n      %s\n      %s" % (name[:100], w, r))
    return 1 if bad or bad_dpp else 0


if __name__ == "__main__":
    sys.exit(main())
