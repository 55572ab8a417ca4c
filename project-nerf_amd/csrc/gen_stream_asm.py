"""Generates mlp_stream_asm.h: the decoder chain kernels' inner pass as ONE inline-asm statement per
256-sample tile and wave -- forward inference (`fwd_stream_pass`), forward with training stash
(`fwd_train_stream_pass`) and the dgrad chain (`bwd_stream_pass`).

Why one statement: hipcc cannot be told to keep MFMAs back to back across m-tile epilogues, to keep
fragment reads in flight across a ring hand-over, to spread stash stores over the MFMA gaps, or to
leave a counted `s_waitcnt` alone.  Inside a single statement every wait state and counter is ours:

  * literal VGPRs (listed as clobbers; the compiler keeps v0..v87): accumulator tiles
    X = v[96:111], Y = v[112:127] alternate per m-tile group; activation buffers P = v[128:191],
    Q = v[192:255] (16 k-steps x 4 registers) alternate per layer; v88..v95 scratch.
  * MFMA g.k issues back to back; the epilogue of group g-1 (v_cvt_pk_bf16_f32 straight into the
    next layer's B operand, ReLU = v_pk_max_i16 on the packed sign bits, mask word, stash stores)
    sits in the MFMA gaps of group g, one unit per gap, starting after the 4th MFMA (the previous
    tile's last MFMA has then left the pipe).
  * forward: the bias of group g+1 is read from LDS straight into the tile the epilogue just
    drained, so every group's first MFMA accumulates in place; dgrad: first MFMA takes C = 0.
  * A fragments: ds_read_b128 D fragments ahead through a rotating VGPR window, continuing across
    groups and chunks; every wait is an exact `s_waitcnt lgkmcnt(n)` / `vmcnt(n)` from a
    simulation of the two in-order queues (stash stores stay in flight across ring hand-overs).
  * ring hand-over of chunk c: B1 = `vmcnt(n) + s_barrier` before the first read of chunk c+1
    (every wave's LDS-DMA of chunk c+1 has landed); B2 = `s_barrier` after the first MFMA of
    chunk c+1 (every wave's reads of chunk c have returned), then the DMA of chunk c+2 into the
    freed slot, one 1-KiB piece per MFMA gap.
  * training: ReLU mask word of tile m = sum_q min(bf16 pair q, 1) << q  (bit q: row 2q active,
    bit 16+q: row 2q+1); its two used bytes go to memory as one 16-bit word per lane and tile (v_perm_b32
    packs / unpacks); dgrad expands it with shift / and 0x10001 / v_pk_mul_lo_u16.
  * training images are 8-bit: the bf16 pair the chain keeps for the next layer is converted once
    more (v_cvt_scalef32_pk_fp8_bf16: e4m3 activations; ..._bf8_bf16: e5m2 pre-activation gradients,
    divided by the power-of-two scale in s95) into the accumulator tile the epilogue has just drained
    (register k of that tile is free once pairs 2k, 2k+1 are converted), so one 16-byte store per lane
    and m-tile carries the tile's 16 rows: block = 1 KiB per (32-sample wave tile, m-tile), lane (c, h)
    at byte 32c + 16h (mlp_chain.h::stash_block8).  MODE.FP16_OVFL makes the conversions saturate.

The chunk tables are recomputed here and pinned against mlp_plan.h by static_asserts in the output.

Run:  python gen_stream_asm.py > mlp_stream_asm.h
"""
import os
import sys
from collections import deque

D = int(os.environ.get("GEN_D", 4))                 # A-fragment prefetch depth (window registers)
D_DEFAULT, D64 = D, int(os.environ.get("GEN_D64", 4))   # infer64 has VGPRs to spare for a deeper window (measured: no effect)
# infer64: epilogue instructions dealt out per fragment.  6 = two whole units (cvt, max, accvgpr_write) per fragment, two instructions
# behind each of the fragment's first three MFMAs: measured best (tools/sweep_infer64.sh: 3 -> -1.5 %, 9 -> -1.7 %, 12+ -> -3.5 % against 6)
S64_MIN = int(os.environ.get("GEN_S64_MIN", 6))
S64_ORDER = int(os.environ.get("GEN_S64_ORDER", 1))      # the two units of a fragment interleaved (cvt cvt | max max | write write): +0.15 %
EPI_START = int(os.environ.get("GEN_EPI", 3))     # first gap (after MFMA k) that may carry epilogue work
# timing ablations (results are wrong with any of these set): GEN_NO=dma,epi,bar,read,store,stinst,vmwait,maskwait,oneimage
ABLATE = set(filter(None, os.environ.get("GEN_NO", "").split(",")))
CHUNK = 64
VA, SO, MO, T0 = 88, 89, 90, 95          # scratch VGPRs; v91..v94: mask words (4 rotating slots in dgrad, v91 forward)
X, Y, P, Q = 96, 112, 128, 192
FIRST_LITERAL_VGPR = 88
# literal SGPRs s84..s99: [84:85] h / dh base, [86:87] feat / dfeat, [88:89] hv / dhv, [90:91] current
# image, [92:93] mask base, [94] 0x00010001, [95] image scale (divisor), [96:97] layer stride in bytes,
# [98:99] scratch, [100] v_perm_b32 selector that packs (forward) / unpacks (dgrad) a mask word
SGPR_LITERALS = range(84, 101)


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def ar(a, n=1):
    return f"a{a}" if n == 1 else f"a[{a}:{a + n - 1}]"


# infer64: 64 samples per wave (four 16-sample halves g = 0..3), four waves per workgroup.  Every A fragment read from LDS feeds
# FOUR MFMAs instead of two: half the fragment reads per sample (11 % of the inference pass on the whole chip, ablation NO=read).
# Registers: accumulator tiles X64 / Y64 = 32 VGPRs each (a(t, g) = T + 4 (4 t + g)); the activation buffers live in AGPRs
# (P64 = a[0:127], Q64 = a[128:255]: [k-step kk][half g][4 registers]) -- MFMA reads its B operand from them directly, the
# epilogue's packed pairs go there through v_accvgpr_write_b32.
X64, Y64, P64, Q64 = 96, 128, 0, 128
BIAS64 = 160      # v[160:167] / v[168:175]: the bias rows of the current / next group (t = 0, 1), the C operand of an accumulator's first MFMA


# ---------------------------------------------------------------- plans
def fwd_groups(s16=False, s64=False):
    """s16: the 16x16x32 MFMA shape (inference).  A group is still one 32-row output tile of a step, now held as
    FOUR 16x16 accumulators a(t, g) = T + 4 (2 t + g): t = 16-row half of the tile, g = 16-sample half of the
    wave's 32 samples.  Fragment k of a group feeds t = k & 1 at k-step kk = k >> 1 (32 deep) with both halves
    g = 0, 1 of the B operand, so every A fragment still feeds 32 cycles of matrix work and the fragment stream
    has exactly the chunking of the 32x32x16 stream."""
    steps = [("PTS0", 8, 0, 4, 0)] + [(f"PTS{l}", 8, 16, 4 if l == 4 else 0, 256 * l) for l in range(1, 8)] + [
        ("HEAD", 9, 16, 0, 2048), ("VIEW", 4, 16, 2, 2048 + 288), ("RGB", 1, 8, 0, 2048 + 288 + 128)]
    Pb, Qb = (P64, Q64) if s64 else (P, Q)
    src = {"PTS0": None, "PTS1": Pb, "PTS2": Qb, "PTS3": Pb, "PTS4": Qb, "PTS5": Pb, "PTS6": Qb, "PTS7": Pb,
           "HEAD": Qb, "VIEW": Pb, "RGB": Qb}
    dst = {"PTS0": Pb, "PTS1": Qb, "PTS2": Pb, "PTS3": Qb, "PTS4": Pb, "PTS5": Qb, "PTS6": Pb, "PTS7": Qb,
           "HEAD": Pb, "VIEW": Qb}
    groups = []
    for li, (name, mt, ks_acc, ks_nat, boff) in enumerate(steps):
        for m in range(mt):
            code = "x" if name in ("PTS0", "PTS4") else "d"
            if s64:
                bops = [tuple(ar(src[name] + 16 * (k >> 1) + 4 * g, 4) for g in range(4)) for k in range(ks_acc)]
                bops += [tuple(f"%[{code}{k >> 1}{'abcd'[g]}]" for g in range(4)) for k in range(ks_nat)]
            elif s16:
                bops = [(vr(src[name] + 4 * (2 * (k >> 1)), 4), vr(src[name] + 4 * (2 * (k >> 1) + 1), 4)) for k in range(ks_acc)]
                bops += [(f"%[{code}{k >> 1}a]", f"%[{code}{k >> 1}b]") for k in range(ks_nat)]
            else:
                bops = [vr(src[name] + 4 * k, 4) for k in range(ks_acc)]
                bops += [f"%[{code}{k}]" for k in range(ks_nat)]
            if name == "RGB":
                epi = dict(kind="rgb")
            elif name == "HEAD" and m == 8:
                epi = dict(kind="sigma")
            else:
                # image: the stash array the tile goes to (training); mask_layer: ReLU mask layer or None
                image = ("h", li) if name.startswith("PTS") else (("feat", 0) if name == "HEAD" else ("hv", 0))
                epi = dict(kind="cvt", dst=dst[name], m=m, relu=name != "HEAD", image=image, mt=4 if name == "VIEW" else 8,
                           mask_layer=None if name == "HEAD" else (li if name.startswith("PTS") else 8))
            groups.append(dict(name=name, m=m, bops=bops, bias=(boff + 32 * m) * 4, epi=epi, src=src[name]))
    return groups


def bwd_groups():
    # (name, m-tiles, k-steps from the previous step, nat operand, dst buffer, src buffer, image, mask layer)
    steps = [("B_RGB", 4, 0, "g0", P, None, ("hv", 0), 8), ("B_VIEW", 8, 8, None, Q, P, ("feat", 0), None),
             ("B_HEAD", 8, 16, "gs", P, Q, ("h", 7), 7)]
    bufs = [(Q, P), (P, Q)]
    for i, l in enumerate(range(7, 0, -1)):            # B_PTS7 .. B_PTS1: d(h_{l-1})
        d, s = bufs[i % 2]
        steps.append((f"B_PTS{l}", 8, 16, None, d, s, ("h", l - 1), l - 1))
    groups = []
    for name, mt, ks_acc, nat, dst, src, image, ml in steps:
        for m in range(mt):
            bops = [vr(src + 4 * k, 4) for k in range(ks_acc)] + ([f"%[{nat}]"] if nat else [])
            epi = dict(kind="cvt", dst=dst, m=m, relu=False, image=image, mt=mt, mask_layer=ml)
            groups.append(dict(name=name, m=m, bops=bops, bias=None, epi=epi, src=src))
    return groups


def chunk_groups(groups):
    chunks, fill = [], CHUNK + 1
    for g in groups:
        ks = len(g["bops"])
        if fill + ks > CHUNK:
            chunks.append(dict(frag0=sum(c["count"] for c in chunks), count=0))
            fill = 0
        g["chunk"], g["off"] = len(chunks) - 1, fill
        fill += ks
        chunks[-1]["count"] += ks
    return chunks


# ---------------------------------------------------------------- stream
def generate(mode):
    img16 = mode in ("train16", "bwd16")          # bf16 training images (two 16-byte stores per lane and m-tile)
    bwd, train, s64 = mode in ("bwd", "bwd16"), mode in ("train", "train16"), mode == "infer64"
    s16 = mode == "infer16" or s64                 # the 16x16x32 shape; s64: four sample halves per wave instead of two
    stash = bwd or train
    groups = bwd_groups() if bwd else fwd_groups(s16, s64)
    WAVES = 4 if s64 else 8                        # waves of a workgroup: each issues 1 KiB of a DMA piece
    NG = 4 if s64 else 2                           # sample halves per wave (16x16x32 shape)
    XT, YT = (X64, Y64) if s64 else (X, Y)
    chunks = chunk_groups(groups)
    n_groups, n_chunks = len(groups), len(chunks)
    assert n_groups % 2 == 0 and n_chunks % 2 == 0
    frags = [(gi, k) for gi, g in enumerate(groups) for k in range(len(g["bops"]))]
    n_frags = len(frags)
    assert all(c["count"] > D + 12 for c in chunks[1:-1]), [c["count"] for c in chunks]

    out, lds_q, vm_q = [], [], []      # emitted lines; LDS / vector-memory operations in issue order
    vm_done = []                       # vector-memory operations of this pass that a wait has already retired
    emit = out.append

    def frag_addr(j):
        gi, k = frags[j]
        g = groups[gi]
        return ("ab1" if g["chunk"] & 1 else "ab0"), (g["off"] + k) * 1024

    def issue_read(j):
        base, off = frag_addr(j)
        if "read" in ABLATE and j >= D:
            return
        emit(f"ds_read_b128 %[w{j % D}], %[{base}] offset:{off}")
        lds_q.append(("w", j))

    def wait_lds(tags):
        hits = [i for i, t in enumerate(lds_q) if t in tags]
        if not hits:
            return                                                  # already covered by an earlier wait
        last = max(hits)
        n = len(lds_q) - 1 - last
        assert n <= 15
        emit(f"s_waitcnt lgkmcnt({n})")
        del lds_q[:last + 1]

    def wait_vm(pred, unknown_ok):
        """all queued vector-memory ops matching pred have completed (in-order return)"""
        hits = [i for i, t in enumerate(vm_q) if pred(t)]
        if not hits:
            if unknown_ok or any(pred(t) for t in vm_done):         # an earlier wait of this pass already covered it
                return
            emit("s_waitcnt vmcnt(0)")                              # issued before this pass: count unknown
            vm_done.extend(vm_q)
            del vm_q[:]
            return
        last = max(hits)
        n = len(vm_q) - 1 - last
        assert n <= 63
        emit(f"s_waitcnt vmcnt({n})")
        vm_done.extend(vm_q[:last + 1])
        del vm_q[:last + 1]

    def dma_piece(cc, p):
        if "dma" in ABLATE:
            return
        c = chunks[cc % n_chunks]
        emit(f"v_add_u32 {vr(VA)}, {hex((c['frag0'] + WAVES * p) * 1024)}, %[voff]")
        emit(f"s_add_u32 m0, %[ldsw], {hex((cc & 1) * CHUNK * 1024 + WAVES * p * 1024)}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 {vr(VA)}, %[src]")
        vm_q.append(("dma", cc))

    def dma_pieces(cc):
        return [(cc, p) for p in range((chunks[cc % n_chunks]["count"] + WAVES - 1) // WAVES)]

    cur_image = [None]

    def set_image(image):
        """SALU: s[90:91] = base of the blocked image the next stores go to"""
        if image == cur_image[0] or ("oneimage" in ABLATE and cur_image[0] is not None):
            return []
        prev, cur_image[0] = cur_image[0], image
        kind, l = image
        if kind == "h":
            if prev is not None and prev[0] == "h" and prev[1] == l - 1:
                return ["s_add_u32 s90, s90, s96", "s_addc_u32 s91, s91, s97"]
            if prev is not None and prev[0] == "h" and prev[1] == l + 1:
                return ["s_sub_u32 s90, s90, s96", "s_subb_u32 s91, s91, s97"]
            if l == 0:
                return ["s_mov_b32 s90, s84", "s_mov_b32 s91, s85"]
            assert l == 7
            return ["s_lshl_b64 s[98:99], s[96:97], 3", "s_sub_u32 s98, s98, s96", "s_subb_u32 s99, s99, s97",
                    "s_add_u32 s90, s84, s98", "s_addc_u32 s91, s85, s99"]
        lo = 86 if kind == "feat" else 88
        return [f"s_mov_b32 s90, s{lo}", f"s_mov_b32 s91, s{lo + 1}"]

    def mask_slot(gi):
        return 91 + (gi % 4 if bwd else 0)

    def mask_load(gi, src="mo0"):
        g = groups[gi % n_groups]
        ml = g["epi"].get("mask_layer")
        if ml is None or "store" in ABLATE:
            return []
        return [f"v_add_u32 {vr(MO)}, {hex((ml * 8 + g['m']) * 1024)}, %[{src}]",
                ("load", f"global_load_ushort {vr(mask_slot(gi))}, {vr(MO)}, s[92:93]", ("ml", gi))]

    def epi_units(gi):
        """gap-sized units of group gi's epilogue (its accumulators are tile T)"""
        g = groups[gi]
        e = g["epi"]
        T = XT if gi % 2 == 0 else YT
        units = []
        if e["kind"] == "sigma":
            if s64:
                return [[f"v_mov_b32 %[sg{g}], {vr(T + 4 * g)}" for g in range(4)]]
            if s16:
                return [[f"v_mov_b32 %[sg0], {vr(T)}", f"v_mov_b32 %[sg1], {vr(T + 4)}"]]
            return [[f"v_mov_b32 %[sg], {vr(T)}"]]
        if e["kind"] != "cvt":
            return []
        r0 = e["dst"] + 8 * e["m"]
        if s64:
            # B operand of k-step m for sample half g = AGPRs r0 + 4g .. +3: [t0 rows 4q..4q+3 | t1 rows 16+4q..]; the packed pair
            # is formed in a scratch VGPR (VALU cannot write AGPRs) and moved over
            r0 = e["dst"] + 16 * e["m"]
            n = 0
            for gsel in range(4):
                for t in range(2):
                    a = T + 4 * (4 * t + gsel)
                    for h in range(2):
                        tmp = 89 + n % 7                              # v89..v95: scratch (v88 is the DMA address)
                        n += 1
                        u = [f"v_cvt_pk_bf16_f32 {vr(tmp)}, {vr(a + 2 * h)}, {vr(a + 2 * h + 1)}"]
                        if e["relu"]:
                            u.append(f"v_pk_max_i16 {vr(tmp)}, {vr(tmp)}, 0")
                        u.append(f"v_accvgpr_write_b32 {ar(r0 + 4 * gsel + 2 * t + h)}, {vr(tmp)}")
                        units.append(u)
            return units
        if s16:
            # B operand of k-step m for sample half g = registers r0 + 4g .. +3: [t0 rows 4q..4q+3 | t1 rows 16+4q..]
            for gsel in range(2):
                for t in range(2):
                    a = T + 4 * (2 * t + gsel)
                    for h in range(2):
                        dstr = r0 + 4 * gsel + 2 * t + h
                        u = [f"v_cvt_pk_bf16_f32 {vr(dstr)}, {vr(a + 2 * h)}, {vr(a + 2 * h + 1)}"]
                        if e["relu"]:
                            u.append(f"v_pk_max_i16 {vr(dstr)}, {vr(dstr)}, 0")
                        units.append(u)
            return units
        masked = e["mask_layer"] is not None
        for j in range(8):
            u = [f"v_cvt_pk_bf16_f32 {vr(r0 + j)}, {vr(T + 2 * j)}, {vr(T + 2 * j + 1)}"]
            if e["relu"]:
                u.append(f"v_pk_max_i16 {vr(r0 + j)}, {vr(r0 + j)}, 0")
            if train and masked and "store" not in ABLATE:
                w = mask_slot(gi)
                if j == 0:
                    u.append(f"v_pk_min_u16 {vr(w)}, {vr(r0)}, s94")
                else:
                    u += [f"v_pk_min_u16 {vr(T0)}, {vr(r0 + j)}, s94", f"v_lshl_or_b32 {vr(w)}, {vr(T0)}, {j}, {vr(w)}"]
            if bwd and masked and "store" not in ABLATE:
                if j == 0:                                           # bytes (even rows, odd rows) -> the halves of a dword
                    u += [("waitmask", gi), f"v_perm_b32 {vr(mask_slot(gi))}, {vr(mask_slot(gi))}, {vr(mask_slot(gi))}, s100"]
                u += [f"v_lshrrev_b32 {vr(T0)}, {j}, {vr(mask_slot(gi))}", f"v_and_b32 {vr(T0)}, s94, {vr(T0)}",
                      f"v_pk_mul_lo_u16 {vr(r0 + j)}, {vr(r0 + j)}, {vr(T0)}"]
            if stash and not img16 and "store" not in ABLATE:
                # 8-bit image: bytes 2j, 2j+1 of the lane's 16-byte record, staged in the drained tile
                cvt8 = "v_cvt_scalef32_pk_bf8_bf16" if bwd else "v_cvt_scalef32_pk_fp8_bf16"
                u.append(f"{cvt8} {vr(T + (j >> 1))}, {vr(r0 + j)}, s95" + (" op_sel:[0,0,1]" if j & 1 else ""))
            units.append(u)
        if stash and "store" not in ABLATE:
            so_base = "so8" if e["mt"] == 8 else "so4"
            if img16:
                # bf16 image: the operand registers themselves, 2-KiB block per (wave tile, m-tile), lane (c, h) at
                # block_lane_offset(c, h) and + 128 (mlp_chain.h::stash_block)
                units.append(set_image(e["image"]) + [f"v_add_u32 {vr(SO)}, {hex(e['m'] * 2048)}, %[{so_base}]",
                                                       ("store", f"global_store_dwordx4 {vr(SO)}, {vr(r0, 4)}, s[90:91] nt")])
                units.append([("store", f"global_store_dwordx4 {vr(SO)}, {vr(r0 + 4, 4)}, s[90:91] offset:128 nt")])
            else:
                units.append(set_image(e["image"]) + [f"v_add_u32 {vr(SO)}, {hex(e['m'] * 1024)}, %[{so_base}]",
                                                       ("store", f"global_store_dwordx4 {vr(SO)}, {vr(T, 4)}, s[90:91] nt")])
            if train and masked:
                units.append([f"v_perm_b32 {vr(mask_slot(gi))}, {vr(mask_slot(gi))}, {vr(mask_slot(gi))}, s100",
                              f"v_add_u32 {vr(MO)}, {hex((e['mask_layer'] * 8 + e['m']) * 1024)}, %[mo0]",
                              ("store", f"global_store_short {vr(MO)}, {vr(mask_slot(gi))}, s[92:93]")])
            if bwd and gi + 4 < n_groups:
                units.append(mask_load(gi + 4))                      # slot gi % 4 is free again
        return units

    def emit_unit(u):
        for line in u:
            if isinstance(line, tuple) and line[0] == "store":
                if "stinst" in ABLATE:            # keep the conversions and mask arithmetic, drop the store instructions
                    continue
                emit(line[1])
                vm_q.append(("st",))
            elif isinstance(line, tuple) and line[0] == "load":
                emit(line[1])
                vm_q.append(line[2])
            elif isinstance(line, tuple) and line[0] == "waitmask":
                if "maskwait" in ABLATE:          # do not wait for the mask word (in-order vmcnt: the wait also covers older stores)
                    continue
                wait_vm(lambda t, gi=line[1]: t == ("ml", gi), unknown_ok=False)
            else:
                emit(line)

    def bias_reads(gi_next, T):
        g = groups[gi_next]
        if g["bias"] is None:
            return []
        if s64:   # bias rows 16 t + 4 q .. +3 (bb carries 16 q) into the group's bias registers: all four sample halves of a tile
            #           start from them (C operand of the first MFMA into each accumulator) -- two reads instead of eight
            Bn = BIAS64 + 8 * (gi_next % 2)
            return [(f"ds_read_b128 {vr(Bn + 4 * t, 4)}, %[bb] offset:{g['bias'] + 64 * t}", ("b", gi_next, t)) for t in range(2)]
        if s16:   # a(t, g) <- bias rows 16 t + 4 q .. +3 (bb carries 16 q); both sample halves start from the same bias
            return [(f"ds_read_b128 {vr(T + 4 * q, 4)}, %[bb] offset:{g['bias'] + 64 * (q >> 1)}", ("b", gi_next, q)) for q in range(4)]
        return [(f"ds_read_b128 {vr(T + 4 * q, 4)}, %[bb] offset:{g['bias'] + 32 * q}", ("b", gi_next, q)) for q in range(4)]

    # ---- pass prologue ----
    emit("s_mov_b32 %[m0s], m0")
    if stash:
        ka = (80, 88, 96, 112, 64) if train else (88, 80, 72, 56, 48)          # h, feat, hv, mask, n_pad
        for sreg, off in zip((84, 86, 88, 92, 96), ka):
            emit(f"s_load_dwordx2 s[{sreg}:{sreg + 1}], %[karg], {hex(off)}")
        emit("s_mov_b32 s94, 0x10001")
        # mask word in registers: bit q = row 2q, bit 16+q = row 2q+1 (q < 8); in memory: 16 bits, byte 0 = even rows,
        # byte 1 = odd rows.  v_perm_b32 selector bytes: 0..3 pick a byte of the source, 0x0c writes zero
        emit("s_mov_b32 s100, " + ("0x0c010c00" if bwd else "0x0c0c0200"))
        if not img16:
            emit("s_mov_b32 s95, %[scale]")                          # divisor of the 8-bit images
            emit("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1")  # MODE.FP16_OVFL: 8-bit conversions saturate
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"s_lshl_b64 s[96:97], s[96:97], {9 if img16 else 8}")  # layer stride = n_pad * 256 elements
    if bwd and "store" not in ABLATE:
        emit("s_cmp_eq_u32 %[first], 0")
        emit("s_cbranch_scc1 .Lwarm%=")
        for gi in range(3):
            emit_unit(mask_load(gi))
        emit(".Lwarm%=:")
        del vm_q[:]                                                  # conditional: treat as issued before the pass
    for line, tag in bias_reads(0, XT):
        emit(line)
        lds_q.append(tag)
    for j in range(min(D, n_frags)):
        issue_read(j)

    pending_dma = deque()
    for gi, g in enumerate(groups):
        ks = len(g["bops"])
        T = XT if gi % 2 == 0 else YT
        Tprev = YT if gi % 2 == 0 else XT
        j0 = sum(len(x["bops"]) for x in groups[:gi])
        first_of_chunk = g["off"] == 0
        units = epi_units(gi - 1) if gi > 0 else []
        if "epi" in ABLATE:
            units = []
        epi_instrs = [line for u in units for line in u] if s64 else []      # s64: dealt out instruction by instruction
        if s64:
            units = []
        if bwd and gi == 0 and "store" not in ABLATE:
            units = [mask_load(3)]                                   # groups 0..2 were prefetched by the previous pass
        # deadline: this block reads, at k-step kd, what the previous group's epilogue writes
        kd = None
        if gi > 0 and groups[gi - 1]["epi"]["kind"] == "cvt" and groups[gi - 1]["epi"]["dst"] == g["src"]:
            kd = 2 * groups[gi - 1]["epi"]["m"]
        bias_next = bias_reads(gi + 1, Tprev) if gi + 1 < n_groups else []
        bias_done = False
        emit(f"; ---- group {gi}: {g['name']} m={g['m']} chunk {g['chunk']} off {g['off']}")
        for k in range(ks):
            j = j0 + k
            need = {("w", j)}
            if k == 0:
                need |= {("b", gi, q) for q in range(8)} & set(lds_q)
            wait_lds(need)
            c_in = vr(T, 16) if (k > 0 or g["bias"] is not None) else "0"
            if s64:
                # One wave per SIMD: nothing else fills this wave's MFMA gaps, and a 16x16x32 MFMA leaves about two simple
                # instructions' worth of free issue behind it (MI355X_MICROARCH.md) -- so the epilogue of the previous group
                # (16 units of cvt / max / accvgpr_write), then the next group's bias reads, are dealt out evenly over the gaps
                # behind this group's MFMAs, to be done one fragment before the group (or the consumer's deadline) ends.  The
                # previous tile's last MFMA has left the pipe after fragment 0 (64 matrix cycles).  The window refill and the DMA
                # piece follow the fragment's fourth MFMA (a refill between MFMAs that still read the window raced them: round 2).
                fill = []
                flush = k == ks - 1 or (kd is not None and k >= kd - 2)
                if k >= 1 or flush:
                    last = min(ks - 2, kd - 3) if kd is not None else ks - 2
                    per_frag = max(S64_MIN, -(-len(epi_instrs) // max(1, last - k + 1)))
                    take = len(epi_instrs) if flush else min(per_frag, len(epi_instrs))
                    fill += [epi_instrs.pop(0) for _ in range(take)]
                    if S64_ORDER and take == 6 and all(isinstance(x, str) for x in fill) and fill[0].startswith("v_cvt") and fill[3].startswith("v_cvt"):
                        fill = [fill[0], fill[3], fill[1], fill[4], fill[2], fill[5]]      # two units interleaved: cvt cvt | max max | write write
                if k == 0:                         # the NEXT group's bias rows into its own registers: nothing waits on the epilogue
                    fill += [("lds", line, tag) for line, tag in bias_next]
                    bias_done = True
                per_gap = -(-len(fill) // 4)
                for gsel in range(4):
                    acc = vr(T + 4 * (4 * (k & 1) + gsel), 4)
                    c_op = vr(BIAS64 + 8 * (gi % 2) + 4 * k, 4) if k < 2 else acc      # an accumulator's first MFMA starts from the bias rows
                    emit(f"v_mfma_f32_16x16x32_bf16 {acc}, %[w{j % D}], {g['bops'][k][gsel]}, {c_op}")
                    for item in fill[gsel * per_gap:(gsel + 1) * per_gap]:
                        if isinstance(item, tuple):
                            emit(item[1])
                            lds_q.append(item[2])
                        else:
                            emit(item)
            elif s16:
                a0, a1 = vr(T + 4 * (2 * (k & 1)), 4), vr(T + 4 * (2 * (k & 1) + 1), 4)
                emit(f"v_mfma_f32_16x16x32_bf16 {a0}, %[w{j % D}], {g['bops'][k][0]}, {a0}")
                # the window register is refilled only AFTER both MFMAs of the pair have issued: a ds_read placed
                # between them (measured +0.85 %) races the second MFMA's operand fetch whenever another wave holds
                # the matrix pipe longer than the LDS latency -- wrong results on some boxes, found in round 2
                emit(f"v_mfma_f32_16x16x32_bf16 {a1}, %[w{j % D}], {g['bops'][k][1]}, {a1}")
            elif "shape32" in ABLATE:
                # timing probe of the 16x16x32 shape (results are wrong): the same operands feed two half-length MFMAs
                for h in range(2):
                    ch = vr(T + 4 * h, 4) if c_in != "0" else "0"
                    emit(f"v_mfma_f32_16x16x32_bf16 {vr(T + 4 * h, 4)}, %[w{j % D}], {g['bops'][k]}, {ch}")
            else:
                emit(f"v_mfma_f32_32x32x16_bf16 {vr(T, 16)}, %[w{j % D}], {g['bops'][k]}, {c_in}")
            # ---- gap fillers ----
            if first_of_chunk and k == 0 and g["chunk"] > 0:
                emit("s_barrier")                                   # B2 of the previous chunk
                pending_dma.extend(dma_pieces(g["chunk"] + 1))
            jn = j + D
            if jn < n_frags:
                gn = groups[frags[jn][0]]
                if gn["off"] == 0 and frags[jn][1] == 0:             # first read of the next chunk: B1
                    if "vmwait" not in ABLATE:
                        wait_vm(lambda t, cc=gn["chunk"]: t == ("dma", cc), unknown_ok=False)
                    emit("s_barrier")
                issue_read(jn)
            if pending_dma and not (first_of_chunk and k == 0):
                dma_piece(*pending_dma.popleft())
            if s64:
                continue                                             # (epilogue and bias reads were dealt out above)
            if k >= EPI_START or k == ks - 1:
                n_units = 1
                if k == ks - 1 or (kd is not None and k >= kd - 2):
                    n_units = len(units)                             # flush (short block or deadline)
                    if units and k < 7:
                        emit("s_nop 11")                             # short block: the previous tile's last MFMA must have drained
                for _ in range(min(n_units, len(units))):
                    emit_unit(units.pop(0))
                if not units and not bias_done:
                    if kd is not None and k >= kd - 2:
                        emit("s_nop 1")
                    for line, tag in bias_next:
                        emit(line)
                        lds_q.append(tag)
                    bias_done = True
        assert not units and bias_done and not epi_instrs
    while pending_dma:
        dma_piece(*pending_dma.popleft())
    # ---- pass end ----
    T = XT if (n_groups - 1) % 2 == 0 else YT
    emit("s_nop 15")
    emit("s_nop 3")
    if s64:
        for gsel in range(4):
            for c in range(3):
                emit(f"v_mov_b32 %[c{3 * gsel + c}], {vr(T + 4 * gsel + c)}")
    elif s16:
        for c in range(3):
            emit(f"v_mov_b32 %[c{c}], {vr(T + c)}")
            emit(f"v_mov_b32 %[c{3 + c}], {vr(T + 4 + c)}")
    elif not bwd:
        for c, name in enumerate(("cr", "cg", "cb")):
            emit(f"v_mov_b32 %[{name}], {vr(T + c)}")
    elif "epi" not in ABLATE:
        for u in epi_units(n_groups - 1):                            # last tile's epilogue, bare
            emit_unit(u)
        if "store" not in ABLATE:
            for gi in range(3):                                      # look-ahead: the next pass's first mask words
                emit_unit(mask_load(n_groups + gi, "mo0n"))
    wait_vm(lambda t: t == ("dma", n_chunks), unknown_ok=True)
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for cc, p in dma_pieces(n_chunks + 1):
        dma_piece(cc, p)
    emit("s_mov_b32 m0, %[m0s]")
    return groups, chunks, out


SIGS = {
    "infer": ("fwd_stream_pass",
              "unsigned ab0, unsigned ab1, unsigned bb, const bf16x8 (&x)[4], const bf16x8 (&d)[2], const char* src,\n"
              "    unsigned voff, unsigned ldsw, float& sg, float& cr, float& cg, float& cb"),
    "infer16": ("fwd_stream16_pass",
                "unsigned ab0, unsigned ab1, unsigned bb, const bf16x8 (&x)[4], const bf16x8 (&d)[2], const char* src,\n"
                "    unsigned voff, unsigned ldsw, float (&sg)[2], float (&c)[6]"),
    "infer64": ("fwd_stream64_pass",
                "unsigned ab0, unsigned ab1, unsigned bb, const bf16x8 (&x)[8], const bf16x8 (&d)[4], const char* src,\n"
                "    unsigned voff, unsigned ldsw, float (&sg)[4], float (&c)[12]"),
    "train": ("fwd_train_stream_pass",
              "unsigned ab0, unsigned ab1, unsigned bb, const bf16x8 (&x)[4], const bf16x8 (&d)[2], const char* src,\n"
              "    unsigned voff, unsigned ldsw, unsigned so8, unsigned so4, unsigned mo0, const void* karg, float scale,\n"
              "    float& sg, float& cr, float& cg, float& cb"),
    "train16": ("fwd_train16_stream_pass",
                "unsigned ab0, unsigned ab1, unsigned bb, const bf16x8 (&x)[4], const bf16x8 (&d)[2], const char* src,\n"
                "    unsigned voff, unsigned ldsw, unsigned so8, unsigned so4, unsigned mo0, const void* karg, float scale,\n"
                "    float& sg, float& cr, float& cg, float& cb"),
    "bwd16": ("bwd16_stream_pass",
              "unsigned ab0, unsigned ab1, const bf16x8& g0, const bf16x8& gs, const char* src, unsigned voff, unsigned ldsw,\n"
              "    unsigned so8, unsigned so4, unsigned mo0, unsigned mo0n, unsigned first, const void* karg, float scale"),
    "bwd": ("bwd_stream_pass",
            "unsigned ab0, unsigned ab1, const bf16x8& g0, const bf16x8& gs, const char* src, unsigned voff, unsigned ldsw,\n"
            "    unsigned so8, unsigned so4, unsigned mo0, unsigned mo0n, unsigned first, const void* karg, float scale"),
}


def emit_function(mode, p):
    global D
    D = D64 if mode == "infer64" else D_DEFAULT
    groups, chunks, lines = generate(mode)
    if "bar" in ABLATE:
        lines = [l for l in lines if l != "s_barrier"]
    name, sig = SIGS[mode]
    tab = "kBwdChunks" if mode == "bwd" else "kFwdChunks"
    if mode in ("infer", "bwd"):
        p(f"static_assert(plan::{tab}.n_chunks == {len(chunks)} && plan::{tab}.n_groups == {len(groups)}, \"stream plan\");")
        for i, c in enumerate(chunks):
            p(f"static_assert(plan::{tab}.chunk_frag0[{i}] == {c['frag0']} && plan::{tab}.chunk_count[{i}] == {c['count']}, \"stream plan\");")
    p(f"__device__ __forceinline__ void {name}(\n    {sig}) {{")
    p("  bf16x8 " + ", ".join(f"w{i}" for i in range(D)) + ";")
    p("  unsigned m0s;")
    p("  asm volatile(")
    for ln in lines:
        if ln.startswith(";"):
            p(f"      // {ln[2:]}")
        else:
            p(f'      "{ln}\\n\\t"')
    outs = [f'[w{i}] "=&v"(w{i})' for i in range(D)] + ['[m0s] "=&s"(m0s)']
    ins = ['[ab0] "v"(ab0)', '[ab1] "v"(ab1)', '[src] "s"(src)', '[voff] "v"(voff)', '[ldsw] "s"(ldsw)']
    if mode in ("bwd", "bwd16"):
        ins += ['[g0] "v"(g0)', '[gs] "v"(gs)', '[mo0n] "v"(mo0n)', '[first] "s"(first)']
    elif mode == "infer64":
        outs += [f'[sg{i}] "=&v"(sg[{i}])' for i in range(4)] + [f'[c{i}] "=&v"(c[{i}])' for i in range(12)]
        ins += ['[bb] "v"(bb)'] + [f'[x{i >> 2}{"abcd"[i & 3]}] "v"(x[{i}])' for i in range(8)] + [f'[d0{"abcd"[i]}] "v"(d[{i}])' for i in range(4)]
    elif mode == "infer16":
        outs += ['[sg0] "=&v"(sg[0])', '[sg1] "=&v"(sg[1])'] + [f'[c{i}] "=&v"(c[{i}])' for i in range(6)]
        ins += ['[bb] "v"(bb)'] + [f'[x{i >> 1}{"ab"[i & 1]}] "v"(x[{i}])' for i in range(4)] + [f'[d0{"ab"[i]}] "v"(d[{i}])' for i in range(2)]
    else:
        outs += ['[sg] "=&v"(sg)', '[cr] "=&v"(cr)', '[cg] "=&v"(cg)', '[cb] "=&v"(cb)']
        ins += ['[bb] "v"(bb)'] + [f'[x{i}] "v"(x[{i}])' for i in range(4)] + [f'[d{i}] "v"(d[{i}])' for i in range(2)]
    if mode not in ("infer", "infer16", "infer64"):
        ins += ['[so8] "v"(so8)', '[so4] "v"(so4)', '[mo0] "v"(mo0)', '[karg] "s"(karg)', '[scale] "s"(scale)']
    clob = [f'"v{i}"' for i in range(FIRST_LITERAL_VGPR, 256)]
    if mode == "infer64":       # accumulators v96..v159, scratch v88..v95; the activation buffers are all 256 AGPRs
        clob = [f'"v{i}"' for i in range(FIRST_LITERAL_VGPR, BIAS64 + 16)] + [f'"a{i}"' for i in range(256)]
    if mode not in ("infer", "infer16", "infer64"):
        clob += [f'"s{i}"' for i in SGPR_LITERALS]
    p("      : " + ", ".join(outs))
    p("      : " + ", ".join(ins))
    p('      : "memory", "scc", ' + ", ".join(clob) + ");")
    p("}\n")
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    print(f"{mode}: groups {len(groups)} chunks {[c['count'] for c in chunks]} lines {len(lines)} mfma {n_mfma}", file=sys.stderr)


def main():
    p = print
    p("// GENERATED by gen_stream_asm.py -- do not edit.  See that file for the design.")
    p(f"// GEN_CONFIG D={D_DEFAULT} NO={','.join(sorted(ABLATE))}" + (f" EPI={EPI_START}" if EPI_START != 3 else "")
      + (f" D64={D64}" if D64 != 4 else "") + (f" S64MIN={S64_MIN}" if S64_MIN != 6 else "") + (f" S64ORDER={S64_ORDER}" if S64_ORDER != 1 else ""))
    p("#pragma once\n")
    p("namespace nerf {\n")
    p(f"static_assert(plan::kChunkFrags == {CHUNK}, \"stream plan\");")
    p("// ab0/ab1: LDS byte address of ring slot 0/1 + lane*16; bb: LDS address of the bias table + 16*half;")
    p("// voff = wave*1024 + lane*16; ldsw = ring base + wave*1024 (wave-uniform); src = fragment stream;")
    p("// so8/so4 = wave_tile*MT*1024 + block8_lane_offset(col, half) for MT = 8/4 (8-bit images; bf16 images:")
    p("// wave_tile*MT*2048 + block_lane_offset(col, half)); mo0 = (tile*72*512 + tid)*2;")
    p("// karg = kernarg segment.\n")
    for mode in os.environ.get("GEN_MODES", "infer,infer16,infer64,train,bwd,train16,bwd16").split(","):
        emit_function(mode, p)
    p("}  // namespace nerf")


if __name__ == "__main__":
    main()
