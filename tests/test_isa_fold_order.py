"""The in-kernel fold's hand-over, checked in the SHIPPED code object (no GPU needed).

Blocks of a step kernel hand their 17 partial sums to the block that draws a group's last ticket
(pcr_icp.hip: wave_reduce17 / store17 / ticket_draw / ticket_redeem).  The sums travel as device-scope write-through stores
(`global_store_dwordx2 ... sc1`), the ticket is a device-scope `global_atomic_add`.  A workgroup barrier does not wait
for outstanding stores on gfx950, so every storing lane must execute `s_waitcnt vmcnt(0)` after its last such store and
before the barrier that precedes the ticket -- otherwise the last block of a group can fold a slot that still holds the
previous launch's value (Weak 2).  This test disassembles libpcr.so and asserts that ordering in the
instruction stream of every kernel that draws a ticket.
"""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pointcloudregistration_amd", "libpcr.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(tmp_path):
    """Every gfx950 code object inside libpcr.so's .hip_fatbin section (one offload bundle per translation unit)."""
    fat = tmp_path / "fat.bin"
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", LIB, str(tmp_path / "discard.so")])
    blob = fat.read_bytes()
    out = []
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (count,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        at = base + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, at)
            triple = blob[at + 24: at + 24 + tlen].decode()
            at += 24 + tlen
            if "gfx950" in triple and size:
                path = tmp_path / f"co_{base}_{len(out)}.elf"
                path.write_bytes(blob[base + off: base + off + size])
                out.append(path)
    return out


def _kernels(tmp_path):
    """name -> list of instruction lines, for every function of every gfx950 code object"""
    funcs = {}
    for co in _code_objects(tmp_path):
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True,
                              text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                funcs[name] = []
            elif name and line.startswith("\t"):
                funcs[name].append(line.strip().split("//")[0].strip())
    return funcs


def _check_ticket_order(name, insts, need_barrier=True):
    """In program order: after a device-scope partial-sum store (`global_store_dwordx2 ... sc1`) no ticket
    (`global_atomic_add`) may be reached before an `s_waitcnt` that brings vmcnt to 0.  Where the stores and the ticket
    come from DIFFERENT waves of a block -- need_barrier: the first ticket of a 256-lane kernel, which hands in the
    tiles of the block's four waves -- an `s_barrier` must follow that wait as well.  Every other ticket is drawn by
    the wave that stored (the filtered kernel's blocks are single waves; a group's fold is one wave's work), and
    `s_waitcnt` is wave-wide."""
    pending_store = None  # index of the last sc1 store
    waited = barrier_after_wait = True
    tickets = 0
    for i, ins in enumerate(insts):
        op = ins.split()[0] if ins else ""
        if op.startswith("global_store_dwordx2") and " sc1" in ins + " ":
            pending_store = i
            waited = barrier_after_wait = False
        elif op == "s_waitcnt" and re.search(r"vmcnt\(0\)", ins):
            waited = True
        elif op == "s_barrier":
            barrier_after_wait = waited
        elif op.startswith("global_atomic_add") and not op.startswith("global_atomic_add_f"):
            tickets += 1
            where = (f"{name}: ticket atomic at instruction {i} ({ins}) follows the sc1 store at {pending_store} "
                     f"({insts[pending_store] if pending_store is not None else '-'}) without ")
            assert waited, where + "`s_waitcnt vmcnt(0)` in between"
            if need_barrier and tickets == 1:
                assert barrier_after_wait, where + "`s_waitcnt vmcnt(0)` + `s_barrier` in between"
    return tickets


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump"), reason="llvm-objdump not in this image")
def test_every_ticket_is_drawn_after_the_partial_sums_were_acknowledged(tmp_path):
    assert os.path.exists(LIB), "libpcr.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    funcs = _kernels(tmp_path)
    step = {n: f for n, f in funcs.items() if "icp_step_" in n and "kernel" in n}
    assert len(step) >= 10, f"expected the step kernels' instances in the code object, found {sorted(step)}"
    with_tickets = 0
    for name, insts in sorted(step.items()):
        n_store = sum(1 for x in insts if x.startswith("global_store_dwordx2") and " sc1" in x + " ")
        assert n_store >= 2, f"{name}: no device-scope partial-sum stores found -- has the hand-over changed?"
        single_wave_blocks = "icp_step_fast_kernel" in name or "icp_step_batch_kernel" in name
        if single_wave_blocks:
            assert not any(x.split()[0] == "s_barrier" for x in insts if x), f"{name}: expected single-wave blocks (no s_barrier)"
        tickets = _check_ticket_order(name, insts, need_barrier=not single_wave_blocks)
        assert tickets >= 2, f"{name}: expected the group ticket and the final ticket, found {tickets} atomics"
        with_tickets += 1
    # the two filtered instances (the headline path), the batch kernel, the fp64 walk, brute force, the normal-aware one
    for must in ("icp_step_fast_kernelILi4E", "icp_step_fast_kernelILi1E", "icp_step_batch_kernelILi4E",
                 "icp_step_grid_kernelILi4ELi0E", "icp_step_grid_kernelILi4ELi1E", "icp_step_grid_kernelILi4ELi2E", "icp_step_brute_kernelILi4E"):
        assert any(must in n for n in step), f"{must} not found among {sorted(step)}"
    assert with_tickets == len(step)


def test_the_checker_rejects_the_round_2_shape():
    """the shape VERDICT r2 found in the shipped ISA: stores, lgkmcnt wait only, barrier, ticket"""
    bad = ["global_store_dwordx2 v1, v[2:3], s[0:1] sc1", "s_waitcnt lgkmcnt(0)", "s_barrier",
           "global_atomic_add_u32 v4, v5, v6, s[2:3] sc0"]
    with pytest.raises(AssertionError):
        _check_ticket_order("bad", bad)
    good = ["global_store_dwordx2 v1, v[2:3], s[0:1] sc1", "s_waitcnt vmcnt(0)", "s_barrier",
            "global_atomic_add_u32 v4, v5, v6, s[2:3] sc0"]
    assert _check_ticket_order("good", good) == 1
