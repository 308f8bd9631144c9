"""Generated-code check for a hazard hipcc does not guard: a v_mfma_f64_16x16x4_f64 whose destination registers overlap
its A operand. The instruction runs in several passes and reads A while it writes D; with `a[16:23], a[22:23], ...` (seen
in the partition-boundary strip kernel at N = 8 when its accumulators started from a literal 0) sixteen rows of the result
were wrong on the MI355X. The kernels written this round start their accumulators from zeros the compiler cannot see
through (mfma_zero / cmfma_zero), which rules the overlap out; this test compiles the highest-order translation units to
assembly (hipcc cross-compiles without a GPU) and scans every matrix instruction."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "blitzdg_amd", "csrc", "hip")
MFMA = re.compile(r"v_mfma_f64_16x16x4_f64 ([av])\[(\d+):(\d+)\], ([av])\[(\d+):(\d+)\], ([av])\[(\d+):(\d+)\]")


def _overlaps(asm):
    """(kernel, instruction, operand) for every matrix instruction whose destination shares a register with A or B."""
    out, kernel = [], None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN7bdg_dev\w+):", line)
        if m:
            kernel = m.group(1)
        m = MFMA.search(line)
        if not m or kernel is None:
            continue
        dt, d0, d1, at, a0, a1, bt, b0, b1 = m.groups()
        d0, d1, a0, a1, b0, b1 = (int(v) for v in (d0, d1, a0, a1, b0, b1))
        if dt == at and not (a1 < d0 or a0 > d1):
            out.append((kernel, line.strip(), "A"))
        if dt == bt and not (b1 < d0 or b0 > d1):
            out.append((kernel, line.strip(), "B"))
    return out


UNITS = [("sw2d_order.hip", 8), ("sw2d_curved_order.hip", 8), ("sw2d_curved_order.hip", 4), ("sw2d_order.hip", 4)]


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    """Every translation unit the tests below read, compiled to assembly side by side (the slowest takes two minutes):
    (source, order) -> text."""
    out = tmp_path_factory.mktemp("isa")
    procs = {}
    for source, order in UNITS:
        asm = out / f"{source}.{order}.s"
        cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", f"-DBDG_ORDER={order}",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "blitzdg_amd", "csrc", "host"), "-I" + HIP,
               "--cuda-device-only", "-S", os.path.join(HIP, source), "-o", str(asm)]
        procs[(source, order)] = (subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True), asm)
    texts = {}
    for key, (proc, asm) in procs.items():
        _, err = proc.communicate(timeout=1500)
        assert proc.returncode == 0, err[-3000:]
        texts[key] = asm.read_text()
    return texts


@pytest.mark.parametrize("source,order", [("sw2d_order.hip", 8), ("sw2d_curved_order.hip", 8), ("sw2d_curved_order.hip", 4)])
def test_no_matrix_instruction_overwrites_an_operand_it_is_still_reading(assembly, source, order):
    text = assembly[(source, order)]
    found = _overlaps(text)
    assert MFMA.search(text)                                             # the scan does see matrix instructions
    # destination overlapping the A operand: never
    assert not [f for f in found if f[2] == "A"], found[:5]
    # ... and no overlap of any kind in the kernels of this round (state-once schedule, strip kernel, curved RHS); the
    # two-waves-per-SIMD kernels of round 1 let the compiler reuse the first pair of D for B (v[0:7], .., v[0:1], 0),
    # which their parity tests (reference fixtures at N = 8 included) show to be harmless
    new = [f for f in found if "mfma3" in f[0] or "curved" in f[0] or "strip" in f[0]]
    assert not new, new[:5]
    if source == "sw2d_order.hip":
        # A spilled register comes back through a scratch load followed by s_waitcnt vmcnt(0), which drains every
        # prefetch the wave has in flight: the straight-sided, single-domain forms of the state-once kernel (what the
        # benchmark and every affine mesh run) must not spill at the highest order.
        spills = dict(re.findall(r"\.name:\s+(_ZN7bdg_dev23sw2d_stage_mfma3_kernel\w+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)",
                                 text))
        # <N, MODE, HALO = false, NODAL = false, NFILT = false, SYNC>: the three modes, and the LSERK instance with in-kernel stage
        # dependencies that the interior launches of a partitioned run take (round 4)
        plain = {k: int(v) for k, v in spills.items() if re.search(r"ILi8ELi\dELb0ELb0ELb0ELb[01]EE", k)}
        assert len(plain) == 4 and not any(plain.values()), plain
        # the tracer as a second phase of every tile (round 4) prefetches during that phase: no spill there either
        phase = dict(re.findall(r"\.name:\s+(_ZN7bdg_dev26sw2d_stage_mfma3src_kernelILi8ELi\dELb0ELi1ELb1E\w+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text))
        assert len(phase) == 6 and not any(int(v) for v in phase.values()), phase   # three modes x (F' products | pointwise sources)
        # the per-node-geometry and halo forms give up the next-tile prefetch at this order for the same reason
        assert len(spills) >= 9 and max(int(v) for v in spills.values()) <= 16, spills


def test_unrolled_kernels_keep_their_one_scheduling_region(assembly):
    """The one-lane-per-element kernels of N <= 4 are one huge unrolled basic block whose operator entries are scalar
    loads the scheduler streams in as it goes. A run-time branch per node inside that body (a sponge test per momentum
    value, `table ? load : constant` per node) splits the region: every operator entry is hoisted and spilled -- the
    midpoint-RK2 / SSP-RK2 forms at N = 4 spilled 1300-1800 scalar and 280-670 vector registers and ran 4-9 times slower
    than the LSERK form of the same kernel until those tests were moved out of the body. Guard: no instance of these
    kernels spills more than a few dozen vector registers, in any time-stepping mode."""
    text = assembly[("sw2d_order.hip", 4)]
    spills = re.findall(r"\.name:\s+(_ZN7bdg_dev\d+(?:sw2d_stage_affine_kernel|sw2d_stage_vb_unrolled_kernel)\w+)\n(?:.*\n)*?"
                        r"\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(spills) >= 20, len(spills)                       # every mode x physics x tracer instance was seen
    worst = max(spills, key=lambda kv: int(kv[1]))
    assert int(worst[1]) <= 64, worst
