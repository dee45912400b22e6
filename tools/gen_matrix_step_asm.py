#!/usr/bin/env python3
"""Writes expressionmatrix2_amd/csrc/em2_matrix_step_asm.h: the inline-assembly body of one tile step of the
matrix-core walk (fsp4ScanMatrixKernel in csrc/em2_scan_symmetric.hip, fsp4TileMatrixKernel in csrc/em2_scan_sharded.hip; the walk itself: csrc/em2_scan_symmetric_device.h).

    python3 tools/gen_matrix_step_asm.py > expressionmatrix2_amd/csrc/em2_matrix_step_asm.h

One step = the 32 x 64 dot products of one 32-column tile with the wave's 64 rows: 16 k-steps x 2
v_mfma_f32_32x32x64_f8f6f4 (FP4 0 / 1 operands -- see ZERO_ONE below --, f32 accumulate), the column fragments read from LDS through a
four-deep register ring with COUNTED lgkmcnt waits (the compiler's own schedule reuses two registers and waits for
lgkmcnt(0) in front of every second k-step, which exposes the LDS latency 8 times per tile), and -- interleaved with
the MFMAs, two results per k-step -- the test of the PREVIOUS tile's results against min(row bound, column bound): the
step of tile t hides the column tests of tile t-1 under its own matrix instructions.

Everything the step touches sits in fixed physical registers that the compiler never sees as values: inline-asm
operands cannot be indexed by sub-register, a step has more operands than the 30 an asm statement may carry, and with
the tuples pinned by "{v[a:b]}" constraints the register allocator moved other values into them and reloaded 32 row
registers from scratch in front of every step.  So v20..v255 belong to the walk: every asm statement of the walk
lists them as clobbered (no value of the compiler's lives there across a step), the rows are written by
EM2_MATRIX_SET_ROW_FRAGMENT, the results are read by EM2_MATRIX_READ_X / _Y, and the steps take scalar operands only
(LDS base addresses): the per-lane state of the walk -- row bounds, log counts -- lives in LDS, so that the walk's
loop holds no vector value of the compiler's across a step or across the call of the events function
(tools/check_matrix_walk_registers.py checks the compiled code):

    v[128:191]  B operand: rows 0..31 of the wave, k-steps 0..15 (4 registers each)
    v[192:255]  B operand: rows 32..63
    v[64:79] / v[80:95]     accumulator set X (rows 0..31 / 32..63)
    v[96:111] / v[112:127]  accumulator set Y
    v[48:63]    ring of four column fragments (A operand)
    v[40:47]    two buffers of four column bounds (the previous tile's, this lane's half)
    v[20:27]    two buffers of four column terms (of the tile the set under test computes next), v36 their LDS address
    v28 / v29 byte offsets of the lane's next record for accumulator 0 / 1 in the wave's log area; v30 / v31 the record
    (stubs);
    v32 lane, then a threshold; v33 / v34 / v35 LDS addresses (tile, bounds, row state), v37 / v38 row bounds,
    v39 the other threshold

Accumulator layout (32x32 result, columns = A = M, rows = B = N): lane l, register i of a set's first / second
accumulator holds row (l & 31) / 32 + (l & 31) and column 8 * (i >> 2) + 4 * (l >> 5) + (i & 3) of the tile.

Test of register i = 4q + j of accumulator a of the previous tile (bounds and results in the accumulators' unit: -mismatches / 2
with the 0 / 1 operands):
    pass = min(rowBound[a], columnBound[8q + 4 * (l >> 5) + j]) <= dot            (q = the "group" of 8 columns)
A register that passes in some lane branches to its stub behind the body: the passing lanes append a record
{first column of the tile | 2i + a, dot} (8 bytes) to their OWN log in global memory -- one log per lane and accumulator,
so that a log holds the records of one row only (v28 / v29 = the lane's byte offsets into the wave's log area, which the
step returns; they persist from step to step).  That is all a step does about an event:
which side of the pair the record is for (row, column, both), the exact state machine and the inbox are the business
of the replay that follows the walk (csrc/em2_scan_symmetric_device.h), which reads the logs lane-parallel, many records per
lane, instead of a few per step.
The column bounds come from the wave's bound scratch in LDS (32 floats per tile), 16 bytes per group and lane half;
the row bounds from the wave's state block (float rowDot[64], lane l reads [l & 31] and [32 + (l & 31)]).
"""
import sys

SETS = {"X": (64, 80), "Y": (96, 112)}
ROWS = (128, 192)
RING = 48
BOUNDS = 40
LANE, TILE_ADDR, BOUND_ADDR, STATE_ADDR, SCALE, ROW_BOUND0, ROW_BOUND1, THR0 = 32, 33, 34, 35, 36, 37, 38, 39
THR1 = LANE          # the lane id is dead once the addresses are formed
OFFSET, RECORD = 28, 30          # v28 (accumulator 0) and v29 (accumulator 1), v[30:31]
TERMS = 20                       # the 0/1 encoding: two buffers of four column terms (v[20:27]), which travel like the column bounds
TERM_ADDR = SCALE                # ... read from LDS at this address (v36 carried the block scale of the v_mfma_scale form once)
FIRST_OWNED = 20                 # (the terms of the lane's two rows are operands of the steps: two registers of the compiler's)
STEPS = 16


def vreg(base, count=1):
    return "v%d" % base if count == 1 else "v[%d:%d]" % (base, base + count - 1)


class Stream:
    def __init__(self):
        self.lines = []
        self.queue = []          # LDS operations in flight, oldest first (they complete in order)
        self.stubs = []          # (register, accumulator) of the tests, in order

    def emit(self, text):
        self.lines.append(text)

    def lds(self, name, text):
        self.lines.append(text)
        self.queue.append(name)

    def wait_for(self, name):
        if name not in self.queue:
            return
        index = self.queue.index(name)
        self.lines.append("s_waitcnt lgkmcnt(%d)" % (len(self.queue) - index - 1))
        del self.queue[:index + 1]


def prologue(s, o, tile, tests):
    """lane id and the lane's LDS addresses from the scalar bases"""
    s.emit("v_mbcnt_lo_u32_b32 %s, -1, 0" % vreg(LANE))
    s.emit("v_mbcnt_hi_u32_b32 %s, -1, %s" % (vreg(LANE), vreg(LANE)))
    if tile:
        s.emit("v_lshl_add_u32 %s, %s, 4, %s" % (vreg(TILE_ADDR), vreg(LANE), o["tileBase"]))          # + 16 * lane
    if tests:
        if CMPX:
            s.emit("s_mov_b64 %s, exec" % o["save"])
        s.emit("v_and_b32 %s, 31, %s" % (vreg(STATE_ADDR), vreg(LANE)))
        s.emit("v_lshl_add_u32 %s, %s, 2, %s" % (vreg(STATE_ADDR), vreg(STATE_ADDR), o["stateBase"]))     # + 4 * (lane & 31)
        s.emit("v_lshrrev_b32 %s, 5, %s" % (vreg(BOUND_ADDR), vreg(LANE)))
        s.emit("v_lshl_add_u32 %s, %s, 4, %s" % (vreg(BOUND_ADDR), vreg(BOUND_ADDR), o["boundBase"]))     # + 16 * (lane >> 5)
        s.lds("rowBound0", "ds_read_b32 %s, %s" % (vreg(ROW_BOUND0), vreg(STATE_ADDR)))
        s.lds("rowBound1", "ds_read_b32 %s, %s offset:128" % (vreg(ROW_BOUND1), vreg(STATE_ADDR)))
        if TILE_BOUND:
            s.emit("v_mov_b32 %s, %s" % (vreg(BOUND_ADDR), o["boundBase"]))
            s.lds("bounds0", "ds_read_b32 %s, %s" % (vreg(BOUNDS), vreg(BOUND_ADDR)))
        else:
            s.lds("bounds0", "ds_read_b128 %s, %s" % (vreg(BOUNDS, 4), vreg(BOUND_ADDR)))
        if ZERO_ONE and "termBase" in o:
            s.emit("v_lshrrev_b32 %s, 5, %s" % (vreg(TERM_ADDR), vreg(LANE)))
            s.emit("v_lshl_add_u32 %s, %s, 4, %s" % (vreg(TERM_ADDR), vreg(TERM_ADDR), o["termBase"]))       # + 16 * (lane >> 5)
            s.lds("terms0", "ds_read_b128 %s, %s" % (vreg(TERMS, 4), vreg(TERM_ADDR)))


def tests_of(s, o, q, j, k, prev0, prev1):
    """Vector half of the test of register k of the two accumulators: the pass masks go to the scalar pairs
    pass0 / pass1 of parity k & 1; their scalar half (shift_in) follows one k-step later, when the masks have long
    arrived -- a scalar instruction that reads a mask a vector compare has just written stalls the wave for the length
    of the vector pipeline, twice per k-step."""
    bound = BOUNDS if TILE_BOUND else BOUNDS + 4 * (q & 1) + j
    if not TILE_BOUND or k == 0:
        s.emit("v_min_f32 %s, %s, %s" % (vreg(THR0), vreg(ROW_BOUND0), vreg(bound)))
        s.emit("v_min_f32 %s, %s, %s" % (vreg(THR1), vreg(ROW_BOUND1), vreg(bound)))
    if CMPX:
        masked_record(s, o, k, 0, THR0, prev0, True)
        masked_record(s, o, k, 1, THR1, prev1, False)
        return
    s.emit("v_cmp_le_f32_e64 %s, %s, %s" % (o["pass0_%d" % (k & 1)], vreg(THR0), vreg(prev0 + k)))
    s.emit("v_cmp_le_f32_e64 %s, %s, %s" % (o["pass1_%d" % (k & 1)], vreg(THR1), vreg(prev1 + k)))


def shift_in(s, o, k):
    """Scalar half: a register whose test passed in some lane goes through its stub (out of line, behind the body)."""
    if CMPX:
        return
    if STUB == "merged":
        # one check for both accumulators: the stub looks at the two masks itself
        s.emit("s_or_b64 vcc, %s, %s" % (o["pass0_%d" % (k & 1)], o["pass1_%d" % (k & 1)]))
        s.emit("s_cbranch_vccnz L_stub_%d_%%=" % k)
        s.emit("L_back_%d_%%=:" % k)
        s.stubs.append((k, None))
    else:
        for a in range(2):
            s.emit("s_cmp_lg_u64 %s, 0" % o["pass%d_%d" % (a, k & 1)])
            s.emit("s_cbranch_scc1 L_stub_%d_%d_%%=" % (k, a))
            s.emit("L_back_%d_%d_%%=:" % (k, a))
            s.stubs.append((k, a))
    if ZERO_ONE and "termBase" in o and (RESTART == "S" or k == STEPS - 1):
        restart(s, o, k)


def restart(s, o, k):
    """The 0/1 encoding: register k of the set under test, tested and recorded, becomes the start of ITS next tile: row term +
    column term of that tile (the next step's first MFMAs take the set as their C operand).  The terms of a group of four
    registers arrive like the bounds, one group ahead."""
    q, j = k >> 2, k & 3
    if j == 0:
        s.wait_for("terms%d" % q)
    term = TERMS + 4 * (q & 1) + j
    prev0, prev1 = o["prev"]
    s.emit("v_add_f32 %s, %s, %s" % (vreg(prev0 + k), vreg(term), o["rowTerm0"]))
    s.emit("v_add_f32 %s, %s, %s" % (vreg(prev1 + k), vreg(term), o["rowTerm1"]))


def stubs(s, o, prev0, prev1):
    """The lanes in which register k of accumulator a passed append one record to their log in global memory
    (record_and_store)."""
    if CMPX:
        s.emit("v_mov_b32 %s, %s" % (o["count"], vreg(OFFSET)))
        s.emit("v_mov_b32 %s, %s" % (o["count1"], vreg(OFFSET + 1)))
        return
    s.emit("s_branch L_end_%=")

    def record(k, a):
        acc = (prev0, prev1)[a] + k
        if STUB == "empty":          # (measurement only: what the two branches of an event cost by themselves)
            return
        mask = o["pass%d_%d" % (a, k & 1)]
        if STUB == "saveexec":
            s.emit("s_and_saveexec_b64 %s, %s" % (o["save"], mask))
        else:
            s.emit("s_mov_b64 %s, exec" % o["save"])
            s.emit("s_mov_b64 exec, %s" % mask)
        record_and_store(s, o, k, a, acc)
        s.emit("s_mov_b64 exec, %s" % o["save"])

    for entry in s.stubs:
        if len(entry) == 3:
            k, a, mfma = entry
            s.emit("L_stub_%d_%d_%%=:" % (k, a))
            s.emit(mfma)
            record(k, a)
            if a == 0:
                # (the other accumulator's check was skipped with the body's)
                s.emit("s_cmp_lg_u64 %s, 0" % o["pass1_%d" % (k & 1)])
                s.emit("s_cbranch_scc0 L_after_%d_%%=" % k)
                record(k, 1)
            s.emit("s_branch L_after_%d_%%=" % k)
            continue
        k, a = entry
        if a is None:
            s.emit("L_stub_%d_%%=:" % k)
            s.emit("s_mov_b64 %s, exec" % o["save"])
            for b in range(2):
                s.emit("s_mov_b64 exec, %s" % o["pass%d_%d" % (b, k & 1)])
                s.emit("s_cbranch_execz L_skip_%d_%d_%%=" % (k, b))
                record_and_store(s, o, k, b, (prev0, prev1)[b] + k)
                s.emit("L_skip_%d_%d_%%=:" % (k, b))
            s.emit("s_mov_b64 exec, %s" % o["save"])
            s.emit("s_branch L_back_%d_%%=" % k)
            continue
        s.emit("L_stub_%d_%d_%%=:" % (k, a))
        record(k, a)
        s.emit("s_branch L_back_%d_%d_%%=" % (k, a))
    s.emit("L_end_%=:")
    s.emit("v_mov_b32 %s, %s" % (o["count"], vreg(OFFSET)))
    s.emit("v_mov_b32 %s, %s" % (o["count1"], vreg(OFFSET + 1)))


import os
# Where a k-step's other work goes: the first group between its two MFMAs, the second behind them.  S = scalar half of
# the previous register's test, M = the two v_min, C = the two v_cmp (after M), D = the fragment read for four k-steps on
# (behind the second MFMA, which reads the slot it refills) and the bound reads.  The second MFMA cannot issue before the
# first has left the pipe, and everything behind it waits with it; spreading the work over both shadows is worth 5 % on
# the step with records (tools/ubench_matrix_step.hip: ",DSMC" 57.1 ms, "M,DCS" 57.4, "SMC,D" 55.7, "S,DMC" 54.5,
# "SM,DC" 54.1-54.9, "MC,DS" 53.6).  EM2_GEN_PLACE overrides for such experiments.
PLACE = os.environ.get("EM2_GEN_PLACE", "SM,DC").split(",")


def fragment_read(s, k, slot, carry):
    """The ring's refill behind k-step k: fragment k + 4 of this tile -- or, in a step that hands the ring over to the step
    of the next tile (carry: the first tile of a pair; its partner sits 16 KB further in LDS), fragment k - 12 of that."""
    if k + 4 < STEPS:
        s.lds("a%d" % (k + 4), "ds_read_b128 %s, %s offset:%d" % (vreg(slot, 4), vreg(TILE_ADDR), 1024 * (k + 4)))
    elif carry:
        s.lds("n%d" % (k + 4 - STEPS), "ds_read_b128 %s, %s offset:%d"
              % (vreg(slot, 4), vreg(TILE_ADDR), 16384 + 1024 * (k + 4 - STEPS)))


def place(s, o, k, what, prev0, prev1, slot, carry=False):
    q, j = k >> 2, k & 3
    bound = BOUNDS if TILE_BOUND else BOUNDS + 4 * (q & 1) + j
    for letter in what:
        if letter == "S" and k:
            shift_in(s, o, k - 1)
        if letter == "M" and (not TILE_BOUND or k == 0):
            if j == 0:
                s.wait_for("bounds%d" % q)
            s.emit("v_min_f32 %s, %s, %s" % (vreg(THR0), vreg(ROW_BOUND0), vreg(bound)))
            s.emit("v_min_f32 %s, %s, %s" % (vreg(THR1), vreg(ROW_BOUND1), vreg(bound)))
        if letter == "D":
            fragment_read(s, k, slot, carry)
            if j == 1 and q < 3 and not TILE_BOUND:
                s.lds("bounds%d" % (q + 1), "ds_read_b128 %s, %s offset:%d"
                      % (vreg(BOUNDS + 4 * ((q + 1) & 1), 4), vreg(BOUND_ADDR), 32 * (q + 1)))
                if ZERO_ONE and "termBase" in o:
                    # (the buffer held group q - 1, whose last register restarted in front of this k-step's first test)
                    s.lds("terms%d" % (q + 1), "ds_read_b128 %s, %s offset:%d"
                          % (vreg(TERMS + 4 * ((q + 1) & 1), 4), vreg(TERM_ADDR), 32 * (q + 1)))
        if letter == "C" and CMPX:
            masked_record(s, o, k, 0, THR0, prev0, True)
            masked_record(s, o, k, 1, THR1, prev1, False)
        elif letter == "C":
            s.emit("v_cmp_le_f32_e64 %s, %s, %s" % (o["pass0_%d" % (k & 1)], vreg(THR0), vreg(prev0 + k)))
            s.emit("v_cmp_le_f32_e64 %s, %s, %s" % (o["pass1_%d" % (k & 1)], vreg(THR1), vreg(prev1 + k)))
            if ZERO_ONE and "termBase" in o and RESTART == "C" and k and "S" in PLACE[0]:
                restart(s, o, k - 1)          # (its stubs ran in front of this k-step's second MFMA)


# The ring of fragments runs through a PAIR of tiles: the step of a pair's first tile (X) ends by reading the first four
# fragments of the second (16 KB further in LDS: the slots of a pair are adjacent and both were staged before the pair's
# barrier), and the step of the second tile (Y) starts with them in flight -- no LDS round trip in front of its first MFMA.
# EM2_GEN_CARRY=0 generates the steps without it (every step fills the ring itself).
CARRY = os.environ.get("EM2_GEN_CARRY", "1") != "0"
# EM2_GEN_TILE_BOUND=1: ONE column bound per tile (the loosest of its 32 columns, a float at boundBase): the two v_min of a
# step's first k-step serve all 32 results, the per-result work is the v_cmp alone.
TILE_BOUND = os.environ.get("EM2_GEN_TILE_BOUND", "0") == "1"
# EM2_GEN_CMPX=1: no branches -- the compare writes EXEC, the record is formed and stored under that mask (nothing happens where
# nothing passed), EXEC is restored; five more instructions per result, all of them masked off almost always (DESIGN 3.1.6:
# 4.9 PFLOP/s at every record rate against 6.6 at the scan's -- instructions under an empty EXEC are not free, the stores least).
CMPX = os.environ.get("EM2_GEN_CMPX", "0") == "1"
# The operand encoding.  "01" (the product since round 6): a signature bit is FP4 0 / 1 on both sides, the dot product is
# popcount(a & b) and mismatches = pa + pb - 2 dot -- so an accumulator STARTS at -(pa_row + pb_column) / 2 and ends at
# -mismatches / 2: bounds, tests and records are in that unit.  Every register of the set under test restarts right behind its
# test and its stubs (restart(): row term + column term of the set's next tile), and a tile's first MFMAs take the set as their
# C operand.  The matrix pipe draws less power on these operands than on +-1 (fewer products that change sign): the step
# holds a higher clock (profiles/r06_scan_experiments.md, 4).  "pm1": the +-1 form of rounds 2-5 (accumulators start at 0, the
# dot product is 1024 - 2 mismatches); tools/ubench_matrix_step.hip can still be built with it.  The 2048-bit (WIDE) steps are
# +-1 in either case.
ZERO_ONE = os.environ.get("EM2_GEN_ENCODING", "01") == "01"
# where a register restarts: "C" (the product) behind the compares of the k-step that follows its own -- its stubs ran in front of
# that k-step's second MFMA --, "S" right behind its stubs, between that k-step's two MFMAs (the microbenchmark: 50.0 against
# 48.7 ms at the bench's record rate; profiles/r06_scan_experiments.md)
RESTART = os.environ.get("EM2_GEN_RESTART", "C")
# EM2_GEN_STUB: the form of the stubs.  "branches" = round 2's (two s_mov around the record); "saveexec" = s_and_saveexec_b64
# instead of the first two; "empty" = no record at all (measurement: the branches alone); "merged" = one scalar test per pair of
# results, the stub looks at both masks; "mfma" = the checks directly in front of the k-step's second MFMA, a stub starts with a
# copy of it (both measured on the 0/1 step in round 6: profiles/r06_scan_experiments.md, 4 -- neither moves the product).  The other forms round 5 measured
# (profiles/r05_scan_experiments.md: the store first / last / narrower / to LDS, one pending record per lane, the k-step's second
# matrix instruction issued from inside the stub) are in the git history: none of them moved the product.
STUB = os.environ.get("EM2_GEN_STUB", "branches")
if STUB == "mfma":
    THR1 = STATE_ADDR          # (a stub runs between a register's v_min and its v_cmp there, and v32 is the record's third word)


def column_bound_register(k):
    """The register that holds the column bound register k of an accumulator was tested against: the buffers of four bounds
    alternate with the groups of four registers, and the buffer of group q is refilled (for group q + 2) only behind the stubs of
    its last register."""
    q, j = k >> 2, k & 3
    return BOUNDS if TILE_BOUND else BOUNDS + 4 * (q & 1) + j


RECORD_BYTES = 16


def record_and_store(s, o, k, a, acc):
    """The record of register k of accumulator a, in the lanes of EXEC: {tile's first column | 2k + a, dot as it stands in the
    accumulator, the column bound it was tested against} -- 12 bytes, 16 apart -- appended to the lane's log.  The third word
    is what lets the replay decide the column side of the pair without a load of its own per record (the column's published
    cut-off as the walk staged it, two pairs of tiles old at most).  v32 (the other threshold of the tests, THR1) is scratch
    inside a stub: the v_min that writes it for the next register comes behind the stubs of this one."""
    s.emit("v_or_b32_e64 %s, %s, %d" % (vreg(RECORD), o["tileCode"], 2 * k + a))
    s.emit("v_mov_b32 %s, %s" % (vreg(RECORD + 1), vreg(acc)))
    s.emit("v_mov_b32 %s, %s" % (vreg(RECORD + 2), vreg(column_bound_register(k))))
    s.emit("global_store_dwordx3 %s, %s, %s" % (vreg(OFFSET + a), vreg(RECORD, 3), o["logBase"]))
    s.emit("v_add_u32 %s, %d, %s" % (vreg(OFFSET + a), RECORD_BYTES, vreg(OFFSET + a)))


def masked_record(s, o, k, a, thr, acc, first):
    """The branch-free test of register k of accumulator a: EXEC = the lanes that pass, their record stored, EXEC restored.
    (The record's third word takes v32, which is THR1: accumulator 1's threshold is re-formed behind accumulator 0's record.)"""
    if a == 1:
        s.emit("v_min_f32 %s, %s, %s" % (vreg(thr), vreg(ROW_BOUND1), vreg(column_bound_register(k))))
    s.emit("v_cmpx_le_f32_e32 vcc, %s, %s" % (vreg(thr), vreg(acc + k)))
    record_and_store(s, o, k, a, acc + k)
    s.emit("s_mov_b64 exec, %s" % o["save"])


def step(cur, prev, tests, operands):
    """cur / prev: 'X' or 'Y'.  operands: placeholder names -> asm operand text."""
    cur0, cur1 = SETS[cur]
    prev0, prev1 = SETS[prev]
    o = dict(operands, prev=(prev0, prev1))
    s = Stream()
    carry_out = CARRY and cur == "X"
    carry_in = CARRY and cur == "Y"
    if carry_in:
        # (in flight since the previous step; whatever the compiler put between the two steps is younger, and LDS
        # operations complete in order: a wait that leaves this many outstanding has the fragment it needs)
        s.queue.extend("a%d" % k for k in range(4))
    prologue(s, o, True, tests)
    for k in range(4):
        if not carry_in:
            s.lds("a%d" % k, "ds_read_b128 %s, %s offset:%d" % (vreg(RING + 4 * k, 4), vreg(TILE_ADDR), 1024 * k))
    for k in range(STEPS):
        slot = RING + 4 * (k % 4)
        s.wait_for("a%d" % k)
        for a, (acc, rows) in enumerate(((cur0, ROWS[0]), (cur1, ROWS[1]))):
            # (the form without block scales: scale 2^0 is what the operands want, and v_mfma_scale_* is two instructions --
            # a v_mfma_ld_scale_b32 in front of this one -- 16 bytes instead of 8 and an issue slot more per MFMA)
            mfma = ("v_mfma_f32_32x32x64_f8f6f4 %s, %s, %s, %s cbsz:4 blgp:4"
                    % (vreg(acc, 16), vreg(slot, 4), vreg(rows + 4 * k, 4), "0" if (k == 0 and not ZERO_ONE) else vreg(acc, 16)))
            if a == 1 and tests and STUB == "mfma" and k:
                # the checks of register k - 1 directly in front of the k-step's second MFMA: a stub starts with a copy of it (the
                # matrix pipe has work while the wave is out of line) and comes back BEHIND the one in the body
                for b in range(2):
                    s.emit("s_cmp_lg_u64 %s, 0" % o["pass%d_%d" % (b, (k - 1) & 1)])
                    s.emit("s_cbranch_scc1 L_stub_%d_%d_%%=" % (k - 1, b))
                    s.stubs.append((k - 1, b, mfma))
            s.emit(mfma)
            if a == 1 and tests and STUB == "mfma" and k:
                s.emit("L_after_%d_%%=:" % (k - 1))
            if a == 0 and tests:
                place(s, o, k, PLACE[0].replace("S", "") if STUB == "mfma" else PLACE[0], prev0, prev1, slot, carry_out)
        if tests:
            place(s, o, k, PLACE[1], prev0, prev1, slot, carry_out)
        else:
            fragment_read(s, k, slot, carry_out)
    # (the next tile's fragments stay in flight: the next step waits for them)
    s.queue = [name for name in s.queue if not name.startswith("n")]
    assert not s.queue, s.queue
    if tests:
        shift_in(s, o, STEPS - 1)
        stubs(s, o, prev0, prev1)
    return s.lines


def init_set(name, operands):
    """The 0/1 encoding: accumulator set `name` becomes the start of a tile -- row term + column term in all 32 registers -- from
    the tile's 32 column terms at termBase (the layout of the bounds).  For the first two tiles of a walk, whose sets no step
    has restarted (and for the pair behind a test without a step).  MFMAs into the set may still be in flight: wait them out."""
    acc0, acc1 = SETS[name]
    o = operands
    s = Stream()
    s.emit("s_nop 15")
    s.emit("s_nop 15")
    s.emit("s_nop 15")
    s.emit("v_mbcnt_lo_u32_b32 %s, -1, 0" % vreg(LANE))
    s.emit("v_mbcnt_hi_u32_b32 %s, -1, %s" % (vreg(LANE), vreg(LANE)))
    s.emit("v_lshrrev_b32 %s, 5, %s" % (vreg(TERM_ADDR), vreg(LANE)))
    s.emit("v_lshl_add_u32 %s, %s, 4, %s" % (vreg(TERM_ADDR), vreg(TERM_ADDR), o["termBase"]))
    homes = (TERMS, TERMS + 4, BOUNDS, BOUNDS + 4)          # (the bounds' buffers are idle outside a step)
    for q in range(4):
        s.lds("terms%d" % q, "ds_read_b128 %s, %s offset:%d" % (vreg(homes[q], 4), vreg(TERM_ADDR), 32 * q))
    for q in range(4):
        s.wait_for("terms%d" % q)
        for j in range(4):
            s.emit("v_add_f32 %s, %s, %s" % (vreg(acc0 + 4 * q + j), vreg(homes[q] + j), o["rowTerm0"]))
            s.emit("v_add_f32 %s, %s, %s" % (vreg(acc1 + 4 * q + j), vreg(homes[q] + j), o["rowTerm1"]))
    assert not s.queue, s.queue
    return s.lines


def test_only(prev, operands):
    """The test of set `prev` without a step: the last tile of a walk.  Its MFMAs may still be in flight and the
    hardware does not interlock a VALU read of an MFMA result: wait them out first."""
    prev0, prev1 = SETS[prev]
    o = operands
    s = Stream()
    s.emit("s_nop 15")
    s.emit("s_nop 15")
    s.emit("s_nop 15")
    prologue(s, o, False, True)
    for q in range(4):
        if q and not TILE_BOUND:
            s.lds("bounds%d" % q, "ds_read_b128 %s, %s offset:%d" % (vreg(BOUNDS + 4 * (q & 1), 4), vreg(BOUND_ADDR), 32 * q))
        s.wait_for("bounds%d" % q)
        for j in range(4):
            if 4 * q + j:
                shift_in(s, o, 4 * q + j - 1)
            tests_of(s, o, q, j, 4 * q + j, prev0, prev1)
    assert not s.queue, s.queue
    shift_in(s, o, STEPS - 1)
    stubs(s, o, prev0, prev1)
    return s.lines


# ---------------------------------------------------------------------------------------------------------------------
# The WIDE step: 2048-bit signatures.  The registers hold 32 rows x 32 k-steps (the rows of ONE accumulator, v[128:255]),
# a tile is 32 columns x 32 k-steps = 32 KB of LDS (two 16 KB slots side by side), a step is 32 k-steps x one MFMA into the
# first accumulator of a set, and a wave walks its columns twice, once per half of its 64 rows.  Everything else is the
# step above: the ring of four fragments, the test of the previous tile's 16 registers (one per two k-steps) against
# min(row bound, column bound), the records {tileCode | 2i, dot} into the log at v28 (the caller puts the half of the
# rows into bit 0 of tileCode, its log's offset into v28 and the half's 32 row bounds at stateBase).
WIDE_STEPS = 32


def wide_prologue(s, o, tile, tests):
    s.emit("v_mbcnt_lo_u32_b32 %s, -1, 0" % vreg(LANE))
    s.emit("v_mbcnt_hi_u32_b32 %s, -1, %s" % (vreg(LANE), vreg(LANE)))
    if tile:
        s.emit("v_lshl_add_u32 %s, %s, 4, %s" % (vreg(TILE_ADDR), vreg(LANE), o["tileBase"]))
    if tests:
        s.emit("v_and_b32 %s, 31, %s" % (vreg(STATE_ADDR), vreg(LANE)))
        s.emit("v_lshl_add_u32 %s, %s, 2, %s" % (vreg(STATE_ADDR), vreg(STATE_ADDR), o["stateBase"]))
        s.emit("v_lshrrev_b32 %s, 5, %s" % (vreg(BOUND_ADDR), vreg(LANE)))
        s.emit("v_lshl_add_u32 %s, %s, 4, %s" % (vreg(BOUND_ADDR), vreg(BOUND_ADDR), o["boundBase"]))
        s.lds("rowBound0", "ds_read_b32 %s, %s" % (vreg(ROW_BOUND0), vreg(STATE_ADDR)))
        if TILE_BOUND:
            s.emit("v_mov_b32 %s, %s" % (vreg(BOUND_ADDR), o["boundBase"]))
            s.lds("bounds0", "ds_read_b32 %s, %s" % (vreg(BOUNDS), vreg(BOUND_ADDR)))
        else:
            s.lds("bounds0", "ds_read_b128 %s, %s" % (vreg(BOUNDS, 4), vreg(BOUND_ADDR)))


def wide_shift_in(s, o, i):
    s.emit("s_cmp_lg_u64 %s, 0" % o["pass0_%d" % (i & 1)])
    s.emit("s_cbranch_scc1 L_stub_%d_0_%%=" % i)
    s.emit("L_back_%d_0_%%=:" % i)
    s.stubs.append((i, 0))


def wide_min(s, i):
    q, j = i >> 2, i & 3
    if TILE_BOUND:
        if i == 0:
            s.wait_for("bounds0")
            s.emit("v_min_f32 %s, %s, %s" % (vreg(THR0), vreg(ROW_BOUND0), vreg(BOUNDS)))
        return
    if j == 0:
        s.wait_for("bounds%d" % q)
    s.emit("v_min_f32 %s, %s, %s" % (vreg(THR0), vreg(ROW_BOUND0), vreg(BOUNDS + 4 * (q & 1) + j)))


def wide_cmp(s, o, i, prev0):
    s.emit("v_cmp_le_f32_e64 %s, %s, %s" % (o["pass0_%d" % (i & 1)], vreg(THR0), vreg(prev0 + i)))


def wide_bounds_ahead(s, i):
    q, j = i >> 2, i & 3
    if j == 1 and q < 3 and not TILE_BOUND:
        s.lds("bounds%d" % (q + 1), "ds_read_b128 %s, %s offset:%d"
              % (vreg(BOUNDS + 4 * ((q + 1) & 1), 4), vreg(BOUND_ADDR), 32 * (q + 1)))


def wide_step(cur, prev, tests, operands):
    cur0 = SETS[cur][0]
    prev0 = SETS[prev][0]
    o = operands
    s = Stream()
    wide_prologue(s, o, True, tests)
    for k in range(4):
        s.lds("a%d" % k, "ds_read_b128 %s, %s offset:%d" % (vreg(RING + 4 * k, 4), vreg(TILE_ADDR), 1024 * k))
    for k in range(WIDE_STEPS):
        slot = RING + 4 * (k % 4)
        i = k >> 1
        s.wait_for("a%d" % k)
        s.emit("v_mfma_f32_32x32x64_f8f6f4 %s, %s, %s, %s cbsz:4 blgp:4"
               % (vreg(cur0, 16), vreg(slot, 4), vreg(ROWS[0] + 4 * k, 4), "0" if k == 0 else vreg(cur0, 16)))
        if k + 4 < WIDE_STEPS:
            s.lds("a%d" % (k + 4), "ds_read_b128 %s, %s offset:%d" % (vreg(slot, 4), vreg(TILE_ADDR), 1024 * (k + 4)))
        if tests:
            if k % 2 == 0:
                if i:
                    wide_shift_in(s, o, i - 1)
                wide_min(s, i)
            else:
                wide_bounds_ahead(s, i)
                wide_cmp(s, o, i, prev0)
    assert not s.queue, s.queue
    if tests:
        wide_shift_in(s, o, 15)
        stubs(s, o, prev0, prev0)
    return s.lines


def wide_test_only(prev, operands):
    prev0 = SETS[prev][0]
    o = operands
    s = Stream()
    s.emit("s_nop 15")
    s.emit("s_nop 15")
    s.emit("s_nop 15")
    wide_prologue(s, o, False, True)
    for i in range(16):
        if i:
            wide_shift_in(s, o, i - 1)
        wide_min(s, i)
        wide_bounds_ahead(s, i)
        wide_cmp(s, o, i, prev0)
    assert not s.queue, s.queue
    wide_shift_in(s, o, 15)
    stubs(s, o, prev0, prev0)
    return s.lines


def c_string(lines, indent="    "):
    return "\n".join('%s"%s\\n"' % (indent, line) for line in lines)


def macro(name, lines):
    return "#define %s \\\n%s\n\n" % (name, " \\\n".join(c_string(lines).split("\n")))


def main():
    out = sys.stdout
    out.write("// em2_matrix_step_asm.h -- GENERATED by tools/gen_matrix_step_asm.py (see there for the register map); do not edit.\n")
    out.write("#ifndef EM2_MATRIX_STEP_ASM_H\n#define EM2_MATRIX_STEP_ASM_H\n\n")
    # Operand order of the asm statements in em2_scan_symmetric_device.h:
    #   step with tests:    %0 / %1 the lanes' record offsets for accumulator 0 / 1 ("=v"), %2..%6 five scratch pairs ("=&s",
    #                       64 bits: the pass masks in flight, the saved exec), then "s": %7 tileBase, %8 boundBase,
    #                       %9 stateBase (LDS byte addresses), %10 logBase (64 bits: the wave's log area), %11 tileCode (first
    #                       column of the tile under test)
    #   step without tests: %0 tileBase
    #   test only:          %0 / %1 record offsets, %2..%6 scratch pairs, %7 boundBase, %8 stateBase, %9 logBase, %10 tileCode
    #   (the 0/1 encoding) step with tests: %12 termBase -- LDS byte address of the 32 column terms of the tile the set under
    #                       test computes next --, %13 / %14 ("v") the terms of the lane's rows l & 31 / 32 + (l & 31);
    #                       EM2_MATRIX_INIT_X / _Y: %0 termBase of the tile the set computes first, %1 / %2 the row terms
    passes = {"pass0_0": "%2", "pass1_0": "%3", "pass0_1": "%4", "pass1_1": "%5", "save": "%6"}
    with_tests = dict(passes, count="%0", count1="%1", tileBase="%7", boundBase="%8", stateBase="%9", logBase="%10", tileCode="%11")
    wide_with_tests = dict(with_tests)
    if ZERO_ONE:
        with_tests.update(termBase="%12", rowTerm0="%13", rowTerm1="%14")
    without = {"tileBase": "%0"}
    only = dict(passes, count="%0", count1="%1", boundBase="%7", stateBase="%8", logBase="%9", tileCode="%10")
    for cur, prev in (("X", "Y"), ("Y", "X")):
        out.write(macro("EM2_MATRIX_STEP_%s_TESTING_%s" % (cur, prev), step(cur, prev, True, with_tests)))
        out.write(macro("EM2_MATRIX_STEP_%s" % cur, step(cur, prev, False, without)))
        out.write(macro("EM2_MATRIX_TEST_%s" % cur, test_only(cur, only)))
        if ZERO_ONE:
            out.write(macro("EM2_MATRIX_INIT_%s" % cur, init_set(cur, {"termBase": "%0", "rowTerm0": "%1", "rowTerm1": "%2"})))
    # the same three for 2048-bit signatures (operand lists as above; %1, %3 and %5 are unused)
    for cur, prev in (("X", "Y"), ("Y", "X")):
        out.write(macro("EM2_MATRIX_WIDE_STEP_%s_TESTING_%s" % (cur, prev), wide_step(cur, prev, True, wide_with_tests)))
        out.write(macro("EM2_MATRIX_WIDE_STEP_%s" % cur, wide_step(cur, prev, False, without)))
        out.write(macro("EM2_MATRIX_WIDE_TEST_%s" % cur, wide_test_only(cur, only)))
    # every vector register the walk owns: no value of the compiler's may live there across any of its asm statements
    owned = ", ".join('"v%d"' % r for r in range(FIRST_OWNED, 256))
    out.write("#define EM2_MATRIX_OWNED_REGISTERS %s\n\n" % owned)
    out.write("#define EM2_MATRIX_STEP_CLOBBERS \"memory\", \"vcc\", \"scc\", EM2_MATRIX_OWNED_REGISTERS\n\n")
    # the lane's record offset lives in a register of the walk from step to step
    out.write("#define EM2_MATRIX_SET_RECORD_OFFSETS \"v_mov_b32 v%d, %%0\\nv_mov_b32 v%d, %%1\\n\"\n\n" % (OFFSET, OFFSET + 1))
    out.write("#define EM2_MATRIX_RECORD_BYTES %du\n\n" % RECORD_BYTES)
    # the 0/1 encoding of the 1024-bit steps: 1 (accumulators in units of -mismatches / 2) or 0 (+-1: 1024 - 2 mismatches)
    out.write("#define EM2_MATRIX_ZERO_ONE %d\n\n" % (1 if ZERO_ONE else 0))
    # the B operand in one go: 32 loads straight into the registers, one wait.  %0 = address of the wave's first row
    # fragment (scalar pair); the fragments of a 32-row block are 1 KB apart (64 lanes x 16 bytes), the second block
    # follows the first
    lines = ["v_mbcnt_lo_u32_b32 %s, -1, 0" % vreg(LANE), "v_mbcnt_hi_u32_b32 %s, -1, %s" % (vreg(LANE), vreg(LANE)),
             "v_lshlrev_b32 %s, 4, %s" % (vreg(TILE_ADDR), vreg(LANE))]
    for index in range(32):
        base = ROWS[index >> 4] + 4 * (index & 15)
        if index and index % 4 == 0:
            lines.append("v_add_u32 %s, 0x1000, %s" % (vreg(TILE_ADDR), vreg(TILE_ADDR)))
        lines.append("global_load_dwordx4 %s, %s, %%0 offset:%d" % (vreg(base, 4), vreg(TILE_ADDR), 1024 * (index % 4)))
    lines.append("s_waitcnt vmcnt(0)")
    out.write(macro("EM2_MATRIX_LOAD_ROWS", lines))
    # the B operand: fragment `index` (0..15 rows 0..31, 16..31 rows 32..63; k-step = index & 15) into its four registers
    out.write("// B operand: fragment index (k-step index & 15 of rows 0..31 for index < 16, of rows 32..63 above) -> its registers\n")
    out.write("#define EM2_MATRIX_SET_ROW_FRAGMENT(index, f) \\\n    switch (index) { \\\n")
    for index in range(32):
        base = ROWS[index >> 4] + 4 * (index & 15)
        text = "\\n".join("v_mov_b32 v%d, %%%d" % (base + c, c) for c in range(4))
        out.write('    case %d: asm volatile("%s" :: "v"((f).x), "v"((f).y), "v"((f).z), "v"((f).w) : EM2_MATRIX_OWNED_REGISTERS); break; \\\n'
                  % (index, text))
    out.write("    default: break; \\\n    }\n\n")
    for name, (acc0, acc1) in SETS.items():
        out.write("// results of register i of accumulator set %s: d0 = rows 0..31, d1 = rows 32..63\n" % name)
        out.write("#define EM2_MATRIX_READ_%s(i, d0, d1) \\\n    switch (i) { \\\n" % name)
        for i in range(16):
            out.write('    case %d: asm volatile("v_mov_b32 %%0, v%d\\nv_mov_b32 %%1, v%d" : "=v"(d0), "=v"(d1)); break; \\\n'
                      % (i, acc0 + i, acc1 + i))
        out.write("    default: __builtin_unreachable(); \\\n    }\n\n")
    out.write("#endif\n")


if __name__ == "__main__":
    main()
