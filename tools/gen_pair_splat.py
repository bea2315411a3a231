"""Writes dmcf_amd/csrc/cconv_pair_splat.inc: the hand-scheduled splat of one 64-pair batch of cconv_pair.hip (splat F).

One v_mfma_f32_4x4x1_16B_f32 adds ONE neighbour pair to the 2 x 2 x 2 cells x 32 channels it touches: block = (z', channel
quad), row i = (y', x'), column j = channel in the quad; the accumulator is the tile of the pair's CLASS (bz, by, bx) -- 27
tiles x 4 registers in v148 .. v255, addressed relative to M0 = 4 * class (s_set_gpr_idx_on, mode src2 | dst).  No ordering,
no padding, no multiply: the A operand (the pair's 8 trilinear products) and the B operand (its feature row) are read from
pair-interleaved staging, one LDS read per GROUP of pairs each (2 pairs with ds_read_b64, 4 with ds_read_b128); the classes of
the batch arrive as 16 scalar registers of four bytes (4 * class) and reach M0 through s_bfe_u32 + s_set_gpr_idx_idx -- scalar
instructions, which issue beside the matrix pipe (a v_readlane_b32 per pair costs 8 clocks of the SIMD: tools/ubench/
pair_splat.hip).  Blocks of 8 pairs: the operands of block b + 1 are fetched between the matrix instructions of block b (fixed
registers v116 .. v147, two buffers).

Hazard the hardware does NOT interlock (same ubench): a matrix instruction whose accumulator was written by the previous
matrix instruction needs two wait states in between -- consecutive pairs may share a class, so every v_mfma here is preceded by
at least two other instructions.

Run from the repository root:  python tools/gen_pair_splat.py
"""
import os

TILE0 = 148                  # first tile register
OPA, OPF = 116, 132          # operand buffers: 16 registers each (2 block parities x 8 pairs)
NBLK = 8

# diagnostic variants for tools/ubench/pair_splat.hip (comma separated flags): b64 = groups of 2 pairs (ds_read_b64), frow =
# features staged row-major (one ds_read_b32 per pair), noreads = no operand reads, noclass = no M0 updates (every pair into
# tile 0); ubench = unpadded record groups
UBENCH_VARIANTS = ["b64,ubench", "frow,ubench", "noclass,ubench"]


def generate(name, variant="", outdir=None):
    gp = 2 if "b64" in variant else 4           # pairs per group
    # staging bytes per group: 8 products / 32 channels x gp pairs (the record groups of the product layout are padded to 36
    # floats: the owner lanes' 4-byte stores then hit 32 different banks)
    a_bytes, f_bytes = (144 if gp == 4 and "ubench" not in variant else 8 * 4 * gp), 32 * 4 * gp
    rd = "ds_read_b64" if gp == 2 else "ds_read_b128"
    # The product stages the features pair-interleaved too ([group][32 channels][4 pairs]: one ds_read_b128 per FOUR pairs;
    # cconv_pair.hip transposes with its LDS stores -- the loads stay 16 bytes per lane); frow: row-major [pair][32 channels], one
    # ds_read_b32 per pair (17.1 instead of 12.7 clocks per pair and SIMD, tools/ubench/pair_splat.hip)
    frow = "frow" in variant

    def opreg(base, pair):  # register holding the operand of `pair` (two buffers of 8)
        return base + (pair & 15)

    def fetch(b):
        out = []
        if "noreads" in variant:
            return out
        for j in range(8 // gp):
            g = (8 * b) // gp + j
            ra, rf = opreg(OPA, 8 * b + gp * j), opreg(OPF, 8 * b + gp * j)
            out.append(f"{rd} v[{ra}:{ra + gp - 1}], %[pa] offset:{g * a_bytes}")
            if frow:
                for t in range(gp):
                    out.append(f"ds_read_b32 v{rf + t}, %[pf] offset:{(8 * b + gp * j + t) * 128}")
            else:
                out.append(f"{rd} v[{rf}:{rf + gp - 1}], %[pf] offset:{g * f_bytes}")
        return out

    out = []
    e = out.append
    for ins in fetch(0):
        e(ins)
    e("s_set_gpr_idx_on %[c0], 0xc")
    for b in range(NBLK):
        nxt = fetch(b + 1) if b + 1 < NBLK else []
        e("s_waitcnt lgkmcnt(0)")
        # where the next block's fetches go: before the pairs that need only one scalar instruction (byte 0 of a class word)
        slots = {0: [], 4: []}
        for i, ins in enumerate(nxt):
            slots[0 if i < len(nxt) // 2 else 4].append(ins)
        for k in range(8):
            pair = 8 * b + k
            m, byte = pair >> 2, pair & 3
            n_between = 0
            if k in slots:
                for ins in slots[k]:
                    e(ins)
                    n_between += 1
            if "noclass" not in variant:
                if byte == 0:
                    e(f"s_set_gpr_idx_idx %[c{m}]")
                    n_between += 1
                else:
                    e(f"s_bfe_u32 %[s0], %[c{m}], {hex((8 << 16) | (8 * byte))}")
                    e("s_set_gpr_idx_idx %[s0]")
                    n_between += 2
            if k == 0:
                n_between += 1  # the s_waitcnt
            if n_between < 2:
                e(f"s_nop {1 - n_between}")
            e(f"v_mfma_f32_4x4x1_16b_f32 v[{TILE0}:{TILE0 + 3}], v{opreg(OPA, pair)}, v{opreg(OPF, pair)}, v[{TILE0}:{TILE0 + 3}]")
        if b + 1 < NBLK:
            e(f"s_cmp_le_u32 %[nb], {b + 1}")
            e("s_cbranch_scc1 Ldone_%=")
    e("Ldone_%=:")
    e("s_set_gpr_idx_off")
    write_inc(name, out, outdir)


PARK0 = OPA  # the second 16 channels of the wave's FIRST point come back from their LDS parking into the (then idle) operand buffers


def tile_reg(bz, by, bx, r):
    return TILE0 + 4 * ((bz * 3 + by) * 3 + bx) + r


def home(bz, y, x):
    """register that ends up holding the sum of everything class plane bz contributes to cell (y, x) of this lane's z'"""
    by, bx = min(y, 2), min(x, 2)
    return tile_reg(bz, by, bx, 2 * (y - by) + (x - bx))


def write_inc(name, lines, outdir=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(outdir or os.path.join(root, "dmcf_amd", "csrc"), name)
    with open(path, "w") as f:
        f.write("// generated by tools/gen_pair_splat.py -- do not edit\n")
        for line in lines:
            f.write('"' + line + '\\n\\t"\n')
    print("wrote", path, len(lines), "lines")


def generate_merge(name, outdir=None):
    """The 27 class tiles of a finished point -> its 64 cells x 32 channels, IN REGISTERS.  A lane holds channel lane & 31 and
    z' = lane >> 5; tile (bz, by, bx) register (y', x') belongs to cell (bz + z', by + y', bx + x').
      1. inside the lane: for every plane class bz the 9 x 4 values fold onto 4 x 4 cells (y, x) -- 20 adds, in place;
      2. planes 1 and 2 are shared by two plane classes held by DIFFERENT half-waves (bz = z, z' = 0 and bz = z - 1, z' = 1):
         one v_permlane32_swap per cell hands the upper half's bz = 0 values down and the lower half's bz = 2 values up, then
         each half adds its own bz = 1 values (row-masked DPP adds: no exec juggling).
    Afterwards lanes 0 .. 31 hold planes 0, 1 and lanes 32 .. 63 planes 2, 3 in home(0, y, x) / home(2, y, x)."""
    out = []
    e = out.append
    e("s_nop 7")  # the tiles were written by matrix instructions: their write -> vector read distance
    e("s_nop 7")
    for bz in range(3):
        for y in range(4):
            for x in range(4):
                h = home(bz, y, x)
                for by in range(3):
                    for bx in range(3):
                        yp, xp = y - by, x - bx
                        if yp in (0, 1) and xp in (0, 1):
                            r = tile_reg(bz, by, bx, 2 * yp + xp)
                            if r != h:
                                e(f"v_add_f32 v{h}, v{h}, v{r}")
    e("s_nop 1")  # vector write -> v_permlane32_swap read: 2 wait states
    for y in range(4):
        for x in range(4):
            e(f"v_permlane32_swap_b32 v{home(0, y, x)}, v{home(2, y, x)}")
    e("s_nop 1")  # vector write -> DPP read: 2 wait states
    for y in range(4):
        for x in range(4):
            h0, h1, h2 = home(0, y, x), home(1, y, x), home(2, y, x)
            e(f"v_add_f32_dpp v{h2}, v{h1}, v{h2} quad_perm:[0,1,2,3] row_mask:0x3 bank_mask:0xf")   # lanes 0 .. 31: plane 1
            e(f"v_add_f32_dpp v{h0}, v{h1}, v{h0} quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf")   # lanes 32 .. 63: plane 2
    write_inc(name, out, outdir)


def generate_zero(name, outdir=None):
    """clear the tiles for the wave's next point (after the LDS stores that read them have completed)"""
    out = ["s_waitcnt lgkmcnt(0)"]
    for r in range(TILE0, TILE0 + 108, 2):
        out.append(f"v_mov_b64 v[{r}:{r + 1}], 0")
    write_inc(name, out, outdir)


def generate_park(name, load, outdir=None):
    """The merged tile of the wave's first point, channels 16 .. 31 (the lanes whose channel belongs to the second chunk), waits
    in the B row of the wave's SECOND point while that point is splatted: value v = (zz, y, x) of parking lane li at
    (v * 32 + li) * 4 bytes -- conflict-free both ways.  load: back into v116 .. v147 ([zz][y][x], the idle operand buffers)."""
    out = []
    for zz in range(2):
        for y in range(4):
            for x in range(4):
                v = 16 * zz + 4 * y + x
                if load:
                    out.append(f"ds_read_b32 v{PARK0 + v}, %[b] offset:{128 * v}")
                else:
                    out.append(f"ds_write_b32 %[b], v{home(2 * zz, y, x)} offset:{128 * v}")
    if load:
        out.append("s_waitcnt lgkmcnt(0)")
    write_inc(name, out, outdir)


def generate_store(name, parked, outdir=None):
    """merged tile -> the point's B row (k' = (z * 4 + y) * 64 + channel * 4 + x): the lane's base address covers its half-wave's
    planes (z = 2 (lane >> 5) + zz) and its channel's column"""
    out = []
    for zz in range(2):
        for y in range(4):
            if parked:
                r = PARK0 + 16 * zz + 4 * y
                out.append(f"ds_write_b128 %[b], v[{r}:{r + 3}] offset:{1024 * zz + 256 * y}")
            else:
                for x in range(4):
                    out.append(f"ds_write_b32 %[b], v{home(2 * zz, y, x)} offset:{1024 * zz + 256 * y + 4 * x}")
    write_inc(name, out, outdir)


def generate_ws_store(name, outdir=None):
    """cconv_ws.hip: the merged tile -> the point's B row with EIGHT 16-byte stores instead of 32 4-byte ones: the 32 sums move
    into the (idle) operand buffers v116 .. v147 first, [zz][y][x] -- a row's four x values are contiguous in k' -- (32 moves:
    cheaper than the 24 LDS instructions they save; the producers of that kernel are alone on their SIMDs)."""
    out = []
    for zz in range(2):
        for y in range(4):
            for x in range(4):
                out.append(f"v_mov_b32 v{PARK0 + 16 * zz + 4 * y + x}, v{home(2 * zz, y, x)}")
    for zz in range(2):
        for y in range(4):
            r = PARK0 + 16 * zz + 4 * y
            out.append(f"ds_write_b128 %[b], v[{r}:{r + 3}] offset:{1024 * zz + 256 * y}")
    write_inc(name, out, outdir)


PRODUCT_FILES = ["cconv_ws_store.inc", "cconv_pair_splat.inc", "cconv_pair_merge.inc", "cconv_pair_zero.inc", "cconv_pair_park.inc",
                 "cconv_pair_unpark.inc", "cconv_pair_store_parked.inc", "cconv_pair_store.inc"]


def write_product(outdir=None):
    """the files cconv_pair.hip includes (tests/test_abi.py regenerates them into a scratch directory and compares)"""
    generate("cconv_pair_splat.inc", "", outdir)
    generate_merge("cconv_pair_merge.inc", outdir)
    generate_zero("cconv_pair_zero.inc", outdir)
    generate_park("cconv_pair_park.inc", False, outdir)
    generate_park("cconv_pair_unpark.inc", True, outdir)
    generate_store("cconv_pair_store_parked.inc", True, outdir)
    generate_store("cconv_pair_store.inc", False, outdir)
    generate_ws_store("cconv_ws_store.inc", outdir)


if __name__ == "__main__":
    write_product()
    ub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench")
    for i, v in enumerate(UBENCH_VARIANTS):
        generate(f"pair_splat_v{i + 1}.inc", v, ub)
