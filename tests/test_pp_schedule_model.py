"""Host-side model of gemm_pp_kernel's operand pipeline (diffsensei_amd/csrc/gemm_pp.hip) - the counted `s_waitcnt vmcnt`
arithmetic of the tile hand-over, checked without a GPU.

One wavefront's vector-memory queue retires in issue order (loads, LDS-DMA pieces and stores alike), so
`s_waitcnt vmcnt(N)` means "everything but the N newest instructions has completed".  The kernel stages eight 16-KiB
half-tiles (A0 A1 B0 B1 x two buffers) with two LDS-DMA instructions each and waits for each one phase before its
fragments are read; between an output tile's prologue and its first k-tile the wave issues the previous tile's C stores
(at least EX_TAIL of them - or EX_TAIL dummy pieces), which the first k-tiles' waits leave in flight.  The model replays
that issue order and asserts, for every fragment read, that the half-tile it reads was (a) staged for exactly this
(tile, k-tile) and (b) covered by a wait executed in an EARLIER phase (the RAW rule of the kernel's header), and for every
restaging that the slot's last read is at least two phases old (one for B0, whose reads are retired before P1's barrier).

The schedule below is a transcription of `stage_prologue`, `ktile` and the loop top; it has to be kept in step with the
kernel by hand - what it buys is that the COUNTS (5 newer half-tiles, EX_TAIL, which waits carry it) are checked as a
whole, including the negative cases: one tail instruction too few, or the extra count on a wait that must not have it.
"""
import pytest

EX_TAIL = 8       # gemm_pp.hip
HT_INSTR = 2      # LDS-DMA instructions per half-tile and wave


class Wave:
    def __init__(self):
        self.queue = []            # issue order: ("dma", slot, tile, kt) | ("tail",)
        self.done = 0              # queue[:done] is guaranteed complete
        self.slot = {}             # slot -> (tile, kt, index of its last instruction, phase issued)
        self.last_read = {}        # slot -> phase of the last fragment read
        self.pending_done = 0      # result of the wait executed in the CURRENT phase (visible from the next one)
        self.phase = 0

    def next_phase(self, n=1):
        self.done = max(self.done, self.pending_done)
        self.phase += n

    def stage(self, op, half, buf, tile, kt):
        s = (op, half, buf)
        if s in self.last_read:    # WAR: the slot's previous content must be out of use
            need = 1 if (op, half) == ("B", 0) else 2
            assert self.phase - self.last_read[s] >= need, f"WAR on {s} in phase {self.phase}"
        for _ in range(HT_INSTR):
            self.queue.append(("dma", s, tile, kt))
        self.slot[s] = (tile, kt, len(self.queue) - 1, self.phase)

    def tail(self, n):
        self.queue += [("tail",)] * n

    def wait(self, newer_half_tiles, ex):
        allowed = HT_INSTR * newer_half_tiles + ex
        self.pending_done = max(self.pending_done, len(self.queue) - allowed)

    def read(self, op, half, buf, tile, kt):
        s = (op, half, buf)
        t, k, idx, _ = self.slot[s]
        assert (t, k) == (tile, kt), f"{s} holds {(t, k)}, wanted {(tile, kt)}"
        assert idx < self.done, f"RAW: {s} of tile {tile} k-tile {kt} read in phase {self.phase} before its wait"
        self.last_read[s] = self.phase


def ktile(w, tile, kt, nk, exa, exb):
    b = kt & 1
    kt1, kt2 = min(kt + 1, nk - 1), min(kt + 2, nk - 1)
    # P1
    w.next_phase()
    w.read("B", 0, b, tile, kt)
    w.read("A", 0, b, tile, kt)
    w.stage("A", 1, b ^ 1, tile, kt1)
    w.wait(5, exa)
    # P2
    w.next_phase()
    w.read("B", 1, b, tile, kt)
    w.stage("B", 0, b, tile, kt2)
    w.wait(5, exb)
    # P3
    w.next_phase()
    w.read("A", 1, b, tile, kt)
    w.stage("A", 0, b, tile, kt2)
    # P4
    w.next_phase()
    w.stage("B", 1, b, tile, kt2)
    w.wait(5, exb)


def prologue(w, tile, nk):
    w.next_phase(2)               # behind the balance barrier: every read of the previous tile has retired
    for op, half in (("A", 0), ("B", 0), ("B", 1), ("A", 1)):
        w.stage(op, half, 0, tile, 0)
    if nk > 1:
        for op, half in (("B", 0), ("A", 0), ("B", 1)):
            w.stage(op, half, 1, tile, 1)


def run(nk, tails, ex_first=(EX_TAIL, EX_TAIL), ex_second=(EX_TAIL, 0), ex_top=EX_TAIL, tiles=3):
    """tails[i]: vector-memory instructions issued between tile i's prologue and its first k-tile."""
    w = Wave()
    prologue(w, 0, nk)
    for t in range(tiles):
        w.tail(tails[t % len(tails)])
        w.wait(5, ex_top)         # loop top
        ktile(w, t, 0, nk, *ex_first)
        ktile(w, t, 1, nk, *ex_second)
        for kt in range(2, nk):
            ktile(w, t, kt, nk, 0, 0)
        if t + 1 < tiles:
            prologue(w, t + 1, nk)   # issued before the epilogue of tile t


@pytest.mark.parametrize("nk", [2, 4, 10, 20, 80])
@pytest.mark.parametrize("tails", [(8,), (9,), (17,), (29,), (8, 17, 9), (11,), (19,), (33,), (11, 33, 19)])
def test_handover_counts_hold(nk, tails):
    """EX_TAIL or more instructions behind every prologue (8 GEGLU stores, 8 pad pieces, +1 bias piece, 16 plain stores
    + bias, 16 stores + 12 residual loads + bias; round 4, fused LayerNorm: + 2 statistics / c pieces on a consumer -
    11 GEGLU, 19 plain -, + 4 statistics stores on a producer - 33): every read is covered, no slot is restaged under a reader."""
    run(nk, tails)


@pytest.mark.parametrize("nk", [2, 4, 20])
def test_one_tail_instruction_too_few_is_an_underwait(nk):
    """The count must never exceed what was issued: with 7 instructions behind the prologue the loop-top wait would let
    A0 / B0 of k-tile 0 be read before they have landed - which is why `pad_tail` exists."""
    with pytest.raises(AssertionError, match="RAW"):
        run(nk, (EX_TAIL - 1,))


def test_extra_count_belongs_to_the_first_five_waits_only():
    """k-tile 1's P2 wait covers A1(k-tile 1), staged AFTER the tail: carrying the extra count there under-waits."""
    with pytest.raises(AssertionError, match="RAW"):
        run(10, (EX_TAIL,), ex_second=(EX_TAIL, EX_TAIL))
    # ... while dropping it anywhere only waits longer (the drained variant, gemm_debug 256, is the limit of that)
    run(10, (17,), ex_first=(0, 0), ex_second=(0, 0), ex_top=0)


def test_five_half_tiles_in_flight_is_the_most_the_reads_allow():
    """Waiting for all but SIX newer half-tiles would read a half-tile that is still in flight."""
    def ktile6(w, tile, kt, nk, exa, exb):
        b = kt & 1
        kt1, kt2 = min(kt + 1, nk - 1), min(kt + 2, nk - 1)
        w.next_phase(); w.read("B", 0, b, tile, kt); w.read("A", 0, b, tile, kt); w.stage("A", 1, b ^ 1, tile, kt1); w.wait(6, exa)
        w.next_phase(); w.read("B", 1, b, tile, kt); w.stage("B", 0, b, tile, kt2); w.wait(6, exb)
        w.next_phase(); w.read("A", 1, b, tile, kt); w.stage("A", 0, b, tile, kt2)
        w.next_phase(); w.stage("B", 1, b, tile, kt2); w.wait(6, exb)
    w = Wave()
    prologue(w, 0, 10)
    w.tail(EX_TAIL)
    w.wait(5, EX_TAIL)
    with pytest.raises(AssertionError, match="RAW"):
        for kt in range(10):
            ktile6(w, 0, kt, 10, 0, 0)
