"""Generates the hand-scheduled gfx950 assembly of the fast sweep's sorting network for TWO independent sets of 64
32-bit elements (one per lane and set): point-cloud-toolbox_amd/csrc/pct_sort_pair.inc.

    python tools/gen_sort_asm.py > point-cloud-toolbox_amd/csrc/pct_sort_pair.inc

The network is the flip form of the bitonic sort (pct_knn.hip: fast_sort_sets): a merge of SIZE elements compares
element i with i ^ (SIZE - 1), then runs strides SIZE/4 .. 1; every compare-exchange leaves the smaller element at
the lower lane, so "keep min or max" is one bit of the lane id (sel0..sel5 = all ones where that bit is set) and a
level is  partner move + v_med3_u32(e, partner, sel).  What the compiler made of the C++ form: one shared temporary
for both sets, so every DPP move waited two states for the previous one (40 s_nop per pair of queries), and each
ds_bpermute was waited for before the other set's was issued.  Here the two sets alternate instruction by
instruction (A.move, A.med3, B.move, B.med3): the two wait states a DPP read needs after the VALU write of its source
are filled by the other set's instructions.  xor-4 levels (no single DPP pattern) use v_min_u32_dpp / v_max_u32_dpp
with bank masks (keep-min lanes are banks 0,2, keep-max lanes banks 1,3): two instructions instead of three.

Hazards tracked (gfx940 family): a DPP / permlane-swap read of a VGPR needs >= 2 wait states after the VALU
write of that VGPR (the destination of a DPP instruction counts as read: it is the `old` operand).
"""
import sys

class Gen:
    def __init__(self):
        self.out = []
        self.since = {}          # register -> instructions issued since its last VALU write

    def emit(self, text, writes=(), dpp_reads=()):
        need = 0
        for r in dpp_reads:
            d = self.since.get(r, 99)
            need = max(need, 2 - d)
        if need > 0:
            self.out.append("s_nop %d" % (need - 1))
            for r in self.since:
                self.since[r] += need
        self.out.append(text)
        if not text.startswith("s_waitcnt"):          # (not relied upon as a wait state)
            for r in self.since:
                self.since[r] += 1
        for r in writes:
            self.since[r] = 0

DPP_CTRL = {1: "quad_perm:[1,0,3,2]", 2: "quad_perm:[2,3,0,1]", 8: "row_ror:8"}
FLIP_CTRL = {2: "quad_perm:[1,0,3,2]", 4: "quad_perm:[3,2,1,0]", 8: "row_half_mirror", 16: "row_mirror"}

def network(sets=("a", "b"), swz=(), duo=False):
    """duo: the two sets are the lower and the upper 64 elements of ONE list of 128 (element = lane + 64 * set): after
    both are sorted, one more merge -- element i against 127 - i (each set against the lane-reversed other one, the
    smaller elements stay in set a), then the half-cleaners of strides 32 .. 1 on either set."""
    g = Gen()
    cur = {s: "e" + s for s in sets}      # register holding the set's elements
    tmp = {s: "t" + s for s in sets}
    for s in sets:
        g.since["%[" + cur[s] + "]"] = 0   # written by the caller's VALU code just before the block
    R = lambda n: "%[" + n + "]"

    def dpp_level(ctrl, sel):
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"v_mov_b32_dpp {t}, {e} {ctrl} row_mask:0xf bank_mask:0xf bound_ctrl:1", writes=[t], dpp_reads=[e, t])
            g.emit(f"v_med3_u32 {e}, {e}, {t}, {R(sel)}", writes=[e])

    def xor4_level():
        # keep-min lanes (lane bit 2 clear: banks 0, 2) take min(e[i], e[i+4]); keep-max lanes (banks 1, 3) max(e[i], e[i-4])
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"v_min_u32_dpp {t}, {e}, {e} row_ror:12 row_mask:0xf bank_mask:0x5", writes=[t], dpp_reads=[e, t])
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"v_max_u32_dpp {t}, {e}, {e} row_ror:4 row_mask:0xf bank_mask:0xa", writes=[t], dpp_reads=[e, t])
        for s in sets:
            cur[s], tmp[s] = tmp[s], cur[s]

    def swap16_level(sel):
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"v_mov_b32 {t}, {e}", writes=[t])
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"v_permlane16_swap_b32 {e}, {t}", writes=[e, t], dpp_reads=[e, t])
            g.emit(f"v_med3_u32 {e}, {e}, {t}, {R(sel)}", writes=[e])

    def swap32_level(sel):
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"v_mov_b32 {t}, {e}", writes=[t])
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"v_permlane32_swap_b32 {e}, {t}", writes=[e, t], dpp_reads=[e, t])
            g.emit(f"v_med3_u32 {e}, {e}, {t}, {R(sel)}", writes=[e])

    def flip128():
        a, b = sets
        ea, ta, eb, tb = R(cur[a]), R(tmp[a]), R(cur[b]), R(tmp[b])
        g.emit(f"ds_bpermute_b32 {tb}, {R('a63')}, {eb}")
        g.emit(f"ds_bpermute_b32 {ta}, {R('a63')}, {ea}")
        g.emit("s_waitcnt lgkmcnt(1)")
        g.emit(f"v_min_u32 {tb}, {ea}, {tb}", writes=[tb])      # lower half: min(a[i], b[63 - i])
        g.emit("s_waitcnt lgkmcnt(0)")
        g.emit(f"v_max_u32 {ta}, {eb}, {ta}", writes=[ta])      # upper half: max(b[i], a[63 - i])
        cur[a], tmp[a], cur[b], tmp[b] = tmp[b], cur[a], tmp[a], cur[b]

    def bperm_level(addr, sel):
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"ds_bpermute_b32 {t}, {R(addr)}, {e}")
        for i, s in enumerate(sets):
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"s_waitcnt lgkmcnt({len(sets) - 1 - i})")
            g.emit(f"v_med3_u32 {e}, {e}, {t}, {R(sel)}", writes=[e])

    def swizzle_level(xor, sel):
        # partner move on the LDS pipe (ds_swizzle, bit mode: lane ^ xor inside 32 lanes): no VALU slot, ~2 LDS cycles
        for s in sets:
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"ds_swizzle_b32 {t}, {e} offset:0x{0x1f | (xor << 10):04x}")
        for i, s in enumerate(sets):
            e, t = R(cur[s]), R(tmp[s])
            g.emit(f"s_waitcnt lgkmcnt({len(sets) - 1 - i})")
            g.emit(f"v_med3_u32 {e}, {e}, {t}, {R(sel)}", writes=[e])

    level_no = [0]

    def stride(st):
        sel = "sel%d" % (st.bit_length() - 1)
        level_no[0] += 1
        if level_no[0] in swz:
            swizzle_level(st, sel)
        elif st == 4:
            xor4_level()
        elif st == 16:
            swap16_level(sel)
        elif st == 32:
            swap32_level(sel)
        else:
            dpp_level(DPP_CTRL[st], sel)

    size = 2
    while size <= 64:
        sel = "sel%d" % (size.bit_length() - 2)
        level_no[0] += 1
        if level_no[0] in swz and size <= 32:
            swizzle_level(size - 1, sel)
        elif size <= 16:
            dpp_level(FLIP_CTRL[size], sel)
        else:
            bperm_level("a31" if size == 32 else "a63", sel)
        st = size // 4
        while st >= 1:
            stride(st)
            st //= 2
        size *= 2
    if duo:
        flip128()
        st = 32
        while st >= 1:
            stride(st)
            st //= 2
    g.out.append("s_nop 1")            # the caller's next instruction may be a DPP read of the result
    return g.out, cur

if __name__ == "__main__":
    # levels (1..21 in network order) whose partner move goes through ds_swizzle instead of DPP / permlane / bpermute
    duo = len(sys.argv) > 1 and sys.argv[1] == "duo"        # python tools/gen_sort_asm.py duo > .../pct_sort_duo.inc
    swz = set(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] and not duo else set()
    lines, cur = network(swz=swz, duo=duo)
    name = "DUO" if duo else "PAIR"
    print("// generated by tools/gen_sort_asm.py %s -- do not edit" % " ".join(sys.argv[1:]))
    print("// result registers: set a in %%[%s], set b in %%[%s]" % (cur["a"], cur["b"]))
    print("#define PCT_SORT_%s_RESULT_A %s" % (name, cur["a"]))
    print("#define PCT_SORT_%s_RESULT_B %s" % (name, cur["b"]))
    print("#define PCT_SORT_%s_ASM \\" % name)
    for l in lines:
        print('    "%s\\n" \\' % l)
    print('    ""')
    n_valu = sum(1 for l in lines if l.startswith("v_"))
    n_nop = sum(1 for l in lines if l.startswith("s_nop"))
    print("// %d VALU, %d s_nop, %d ds_bpermute, %d ds_swizzle" % (n_valu, n_nop, sum(1 for l in lines if l.startswith("ds_bperm")), sum(1 for l in lines if l.startswith("ds_swizzle"))))
