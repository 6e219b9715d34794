"""Static check of the shipped code object: no PACKED VALU instruction (any `v_pk_*`, v_pk_mov_b32 included) may select a HIGH source
half / dword for its LOW lane (`op_sel:` with a 1; `op_sel_hi` forms - broadcasts of a low half - are not matched).  Found in round 2 on MI355X: `v_pk_fma_f32 vD, vA, vS, vT op_sel:[0,1,1]` (emitted by the SLP vectoriser for two adjacent
pixels sharing one scale / shift) intermittently computed its low result with a zero product in lanes 48-63 when it followed the VALU
write of vA.lo - about once per 10^6 executions, only under full-chip load (tools/diag_batch4.py; DESIGN.md "A hardware / toolchain
hazard").  The cause was never established (16 wait states made it 1000x rarer, not gone), so the guard is as wide as the instruction class:
every packed op with a set op_sel bit, in every kernel of the linked library (k_pwr.o, the one object still built with the SLP vectoriser,
included).  The library is built with -fno-slp-vectorize; this check keeps the pattern from coming back.

    python tools/isa_check.py [path/to/librtfs_amd.so]      exit status 1 and a listing when the pattern is present
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PAT = re.compile(r"\bv_pk_\w+\b.*\bop_sel:\[[01,]*1[01,]*\]")


def device_disassembly(so_path):
    """Yield (kernel, instruction) for every instruction of every gfx950 code object bundled in `so_path`."""
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, "lib.so")
        shutil.copy(so_path, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = sorted(f for f in os.listdir(td) if "amdgcn" in f)
        if not cos:
            raise RuntimeError(f"no gfx950 code object found in {so_path}")
        for co in cos:
            out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(td, co)], check=True, stdout=subprocess.PIPE, text=True).stdout
            kernel = "?"
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    kernel = m.group(1)
                elif line.startswith("\t") or line.startswith(" "):
                    yield kernel, line.strip()


def hazardous(so_path):
    return [(k, ins) for k, ins in device_disassembly(so_path) if PAT.search(ins)]


# ---- second check (round 3): a matrix instruction must not read a VGPR that a VALU instruction wrote fewer than MFMA_WAIT wait states earlier.
# The compiler inserts these wait states for the VALU instructions it knows; it does not look inside inline assembly - neither at a hand-written
# `v_mfma` (pipe_helpers.h: every one carries its own `s_nop 1`) nor at an inline-asm PRODUCER such as split2()'s v_fma_mixlo / mixhi in front
# of a compiler-generated `v_mfma`.  Found as stale fragment registers (zeros at random pixels, NOTES.md); the check reads the linked code.
MFMA_WAIT = 2
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(tok):
    out = set()
    for m in _VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def mfma_read_hazards(so_path):
    """(kernel, producer, consumer, wait states) for every v_mfma whose A / B / C source VGPR was last written by a VALU instruction fewer
    than MFMA_WAIT wait states before it (an intervening instruction counts as one wait state, `s_nop N` as N + 1; a label ends the window)."""
    bad, window, kernel = [], [], None  # window: [(text, dest vregs or None), ...] most recent last
    for k, ins in device_disassembly(so_path):
        if k != kernel:
            kernel, window = k, []
        op = ins.split()[0] if ins.split() else ""
        if op.startswith("v_mfma"):
            ops = [t.strip() for t in ins.split(None, 1)[1].split(",")]
            src = set()
            for t in ops[1:4]:
                src |= _vregs(t)
            ws = 0
            for text, dst in reversed(window):
                if ws >= MFMA_WAIT:
                    break
                if dst and (dst & src):
                    bad.append((k, text, ins, ws))
                    break
                m = re.match(r"s_nop\s+(\d+)", text)
                ws += int(m.group(1)) + 1 if m else 1
        dst = None
        if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_accvgpr_write")) and len(ins.split(None, 1)) > 1:
            dst = _vregs(ins.split(None, 1)[1].split(",")[0])  # (v_readlane / v_readfirstlane write SGPRs: no v-register in the first operand)
        window.append((ins, dst))
        if len(window) > 8:
            window.pop(0)
    return bad


# ---- third check: the VGPR result of an 8-pass matrix instruction must not be read OR OVERWRITTEN by a VALU / memory instruction fewer than
# MFMA_RESULT_WAIT wait states behind it.  The compiler pads its own pairs (it gives a VGPR-form builtin 12); pipe_helpers.h's hand-written
# sequences end in mfma_v_fence() (s_nop 15 + s_nop 3) or put a dozen matrix instructions in between - but a hand-written instruction whose
# result is DEAD (the pipelines' work for the non-existent chunk behind the last) is followed by whatever the compiler puts into the freed
# registers: found by this check in tail_s3t_kernel (a `v_lshlrev_b32 v0` nine wait states behind a `v_mfma ... v[0:15]`: the late result
# write would have landed on the new value).  AGPR results are left to the compiler (no hand-written instruction has one).  Counted as the
# compiler counts: one wait state per instruction, N + 1 per `s_nop N`.
MFMA_RESULT_WAIT = 11
_AVREG = re.compile(r"\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]")


def _regs(tok):
    out = set()
    for m in _AVREG.finditer(tok):
        if m.group(1) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def mfma_result_hazards(so_path):
    bad, kernel, pending = [], None, {}  # pending: register -> (wait states since the matrix instruction that wrote it, its text)
    for k, ins in device_disassembly(so_path):
        if k != kernel:
            kernel, pending = k, {}
        parts = ins.split(None, 1)
        op = parts[0] if parts else ""
        m = re.match(r"s_nop\s+(\d+)", ins)
        step = int(m.group(1)) + 1 if m else 1
        if op.startswith(("v_", "ds_", "buffer_", "global_", "flat_", "scratch_")) and not op.startswith("v_mfma") and len(parts) > 1:
            ops = [t.strip() for t in parts[1].split(",")]
            for r in set().union(*[_regs(t) for t in ops]):
                if r in pending and pending[r][0] < MFMA_RESULT_WAIT:
                    bad.append((k, pending[r][1], ins, pending[r][0]))
                    break
        for r in list(pending):
            w, text = pending[r]
            if w + step >= 64:
                del pending[r]
            else:
                pending[r] = (w + step, text)
        if op.startswith("v_mfma") and len(parts) > 1:
            for r in _regs(parts[1].split(",")[0]):
                if r[0] == "v":
                    pending[r] = (0, ins)
    return bad


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rtfs-net_amd", "librtfs_amd.so")
    bad = hazardous(so)
    for k, ins in bad:
        print(f"{k}: {ins}")
    print(f"{len(bad)} packed instruction(s) with a set op_sel bit in {so}")
    bad2 = mfma_read_hazards(so)
    for k, prod, cons, ws in bad2[:40]:
        print(f"{k}: {prod}  ->  {cons}   ({ws} wait state(s))")
    print(f"{len(bad2)} matrix instruction(s) reading a VGPR fewer than {MFMA_WAIT} wait states behind its VALU write in {so}")
    bad3 = mfma_result_hazards(so)
    for k, prod, cons, ws in bad3[:40]:
        print(f"{k}: {prod}  ->  {cons}   ({ws} wait state(s))")
    print(f"{len(bad3)} VALU access(es) to a matrix instruction's VGPR result fewer than {MFMA_RESULT_WAIT} wait states behind it in {so}")
    sys.exit(1 if bad or bad2 or bad3 else 0)
