"""Generates mlp_fwd_stream_asm.h: ONE inline-asm statement that carries a wave's 32 samples through
all 11 GEMM steps of the decoder (inference), software-pipelined across m-tiles and LDS-ring chunks.

Why one statement: hipcc cannot be told to keep MFMAs back to back across m-tile epilogues, to keep
fragment reads in flight across a ring hand-over, or to leave a counted `s_waitcnt` alone.  Inside a
single statement every wait state and every counter is ours:

  * accumulator tiles X = a[0:15], Y = a[16:31] alternate per m-tile group; activation buffers
    P = a[32:95], Q = a[96:159] (16 k-steps x 4 registers) alternate per layer.  All literal AGPRs,
    listed as clobbers, so the compiler never touches them and they persist across tile passes.
  * MFMA g.k issues back to back; the epilogue of group g-1 (v_accvgpr_read, v_cvt_pk_bf16_f32,
    v_pk_max_i16 = ReLU on the packed bf16 sign bits, v_accvgpr_write into the next layer's B
    operand) is placed in the MFMA gaps of group g, EPI_PER_GAP units at a time, starting after
    the 4th MFMA (the previous tile's last MFMA has then left the pipe).
  * the bias of group g+1 is read from LDS straight into the tile the epilogue just drained
    (ds_read_b128 into AGPRs), so the first MFMA of every group accumulates in place.
  * A fragments: ds_read_b128 D fragments ahead through a rotating VGPR window, continuing across
    groups and chunks; every wait is an exact `s_waitcnt lgkmcnt(n)` from a simulation of the LDS
    queue.
  * ring hand-over of chunk c: B1 = `vmcnt(0) + s_barrier` before the first read of chunk c+1
    (every wave's LDS-DMA of chunk c+1 has landed); B2 = `s_barrier` after the first MFMA of
    chunk c+1 (every wave's reads of chunk c have returned), then the DMA of chunk c+2 into the
    freed slot, one 1-KiB piece per MFMA gap.

The chunk table is recomputed here and pinned against mlp_plan.h by static_asserts in the output.

Run:  python gen_fwd_stream_asm.py > mlp_fwd_stream_asm.h
"""
import sys
from collections import deque

import os
D = int(os.environ.get("GEN_D", 6))                 # A-fragment prefetch depth (window registers)
EPI_PER_GAP = int(os.environ.get("GEN_EPI", 1))     # epilogue units (5 VALU ops) per MFMA gap
EPI_START = 3     # first gap (after MFMA k) that may carry epilogue work
# timing ablations (results are wrong with any of these set): GEN_NO=dma,epi,bar,read
ABLATE = set(filter(None, os.environ.get("GEN_NO", "").split(",")))
CHUNK = 64
X, Y, P, Q = 96, 112, 128, 192   # literal VGPRs v96..v255 (clobbered); the compiler keeps v0..v95

# (name, m-tiles, k-steps from the previous layer, k-steps from codes, bias offset in floats)
STEPS = [("PTS0", 8, 0, 4, 0)] + [(f"PTS{l}", 8, 16, 4 if l == 4 else 0, 256 * l) for l in range(1, 8)] + [
    ("HEAD", 9, 16, 0, 2048), ("VIEW", 4, 16, 2, 2048 + 288), ("RGB", 1, 8, 0, 2048 + 288 + 128)]
SRC = {"PTS0": None, "PTS1": P, "PTS2": Q, "PTS3": P, "PTS4": Q, "PTS5": P, "PTS6": Q, "PTS7": P,
       "HEAD": Q, "VIEW": P, "RGB": Q}
DST = {"PTS0": P, "PTS1": Q, "PTS2": P, "PTS3": Q, "PTS4": P, "PTS5": Q, "PTS6": P, "PTS7": Q,
       "HEAD": P, "VIEW": Q, "RGB": None}


def build_groups():
    groups = []
    for name, mt, ks_acc, ks_nat, boff in STEPS:
        for m in range(mt):
            bops = [f"v[{SRC[name] + 4 * k}:{SRC[name] + 4 * k + 3}]" for k in range(ks_acc)]
            code = "x" if name in ("PTS0", "PTS4") else "d"
            bops += [f"%[{code}{k}]" for k in range(ks_nat)]
            if name == "RGB":
                epi = ("rgb",)
            elif name == "HEAD" and m == 8:
                epi = ("sigma",)
            else:
                epi = ("cvt", DST[name], m, name != "HEAD")
            groups.append(dict(name=name, m=m, bops=bops, bias=(boff + 32 * m) * 4, epi=epi, src=SRC[name]))
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


def generate():
    groups = build_groups()
    chunks = chunk_groups(groups)
    n_groups, n_chunks = len(groups), len(chunks)
    assert n_groups % 2 == 0 and n_chunks % 2 == 0
    frags = [(gi, k) for gi, g in enumerate(groups) for k in range(len(g["bops"]))]
    n_frags = len(frags)
    assert all(c["count"] > D + 12 for c in chunks[:-1]), [c["count"] for c in chunks]

    out, lds_q = [], []      # emitted lines; LDS operations in issue order (tags)
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

    def wait_for(tags):
        hits = [i for i, t in enumerate(lds_q) if t in tags]
        if not hits:
            return                                                  # already covered by an earlier wait
        last = max(hits)
        n = len(lds_q) - 1 - last
        assert n <= 15
        emit(f"s_waitcnt lgkmcnt({n})")
        del lds_q[:last + 1]

    def dma_piece(cc, p):
        if "dma" in ABLATE:
            return
        c = chunks[cc % n_chunks]
        emit(f"v_add_u32 %[va], {hex((c['frag0'] + 8 * p) * 1024)}, %[voff]")
        emit(f"s_add_u32 m0, %[ldsw], {hex((cc & 1) * CHUNK * 1024 + 8 * p * 1024)}")
        emit("s_nop 0")
        emit("global_load_lds_dwordx4 %[va], %[src]")

    def dma_pieces(cc):
        return [(cc, p) for p in range((chunks[cc % n_chunks]["count"] + 7) // 8)]

    def epi_units(gi):
        """VALU units of group gi's epilogue (tile T); returns (units, bias_reads)"""
        g = groups[gi]
        T = X if gi % 2 == 0 else Y
        units = []
        if g["epi"][0] == "cvt":
            _, dst, m, relu = g["epi"]
            for j in range(8):
                r = dst + 4 * (2 * m + j // 4) + j % 4
                u = [f"v_cvt_pk_bf16_f32 v{r}, v{T + 2 * j}, v{T + 2 * j + 1}"]
                if relu:
                    u.append(f"v_pk_max_i16 v{r}, v{r}, 0")
                units.append(u)
        elif g["epi"][0] == "sigma":
            units.append([f"v_mov_b32 %[sg], v{T}"])
        return units

    def bias_reads(gi_next, T):
        g = groups[gi_next % n_groups]
        lines = []
        for q in range(4):
            lines.append((f"ds_read_b128 v[{T + 4 * q}:{T + 4 * q + 3}], %[bb] offset:{g['bias'] + 32 * q}", ("b", gi_next, q)))
        return lines

    emit("s_mov_b32 %[m0s], m0")
    for line, tag in bias_reads(0, X):
        emit(line)
        lds_q.append(tag)
    for j in range(min(D, n_frags)):
        issue_read(j)
    pending_dma = deque()
    for gi, g in enumerate(groups):
        ks = len(g["bops"])
        T = X if gi % 2 == 0 else Y
        Tprev = Y if gi % 2 == 0 else X
        j0 = sum(len(x["bops"]) for x in groups[:gi])
        first_of_chunk = g["off"] == 0
        units = epi_units(gi - 1) if gi > 0 else []
        if "epi" in ABLATE:
            units = []
        # deadline: this block reads, at k-step kd, what the previous group's epilogue writes
        kd = None
        if gi > 0 and groups[gi - 1]["epi"][0] == "cvt" and groups[gi - 1]["epi"][1] == g["src"]:
            kd = 2 * groups[gi - 1]["epi"][2]
        bias_next = bias_reads(gi + 1, Tprev) if gi + 1 < n_groups else []
        bias_done = False
        emit(f"; ---- group {gi}: {g['name']} m={g['m']} chunk {g['chunk']} off {g['off']}")
        for k in range(ks):
            j = j0 + k
            need = {("w", j)}
            if k == 0:
                need |= {("b", gi, q) for q in range(4)} & set(lds_q)
            wait_for(need)
            emit(f"v_mfma_f32_32x32x16_bf16 v[{T}:{T + 15}], %[w{j % D}], {g['bops'][k]}, v[{T}:{T + 15}]")
            # ---- gap fillers ----
            if first_of_chunk and k == 0 and g["chunk"] > 0:
                emit("s_barrier")                                   # B2 of the previous chunk
                pending_dma.extend(dma_pieces(g["chunk"] + 1))
            jn = j + D
            if jn < n_frags:
                gn = groups[frags[jn][0]]
                if gn["off"] == 0 and frags[jn][1] == 0:             # first read of the next chunk: B1
                    emit("s_waitcnt vmcnt(0)")
                    emit("s_barrier")
                issue_read(jn)
            if pending_dma and not (first_of_chunk and k == 0):
                dma_piece(*pending_dma.popleft())
            if k >= EPI_START or k == ks - 1:
                n_units = EPI_PER_GAP
                if k == ks - 1 or (kd is not None and k >= kd - 2):
                    n_units = len(units)                             # flush (short block or deadline)
                    if units and k < 7:
                        emit("s_nop 7")                              # short block: let the previous tile's last MFMA drain
                for _ in range(min(n_units, len(units))):
                    for line in units.pop(0):
                        emit(line)
                if not units and not bias_done:
                    if kd is not None and k >= kd - 2:
                        emit("s_nop 1")
                    for line, tag in bias_next:
                        emit(line)
                        lds_q.append(tag)
                    bias_done = True
        assert not units and bias_done
    while pending_dma:
        dma_piece(*pending_dma.popleft())
    # ---- pass end: rgb tile -> VGPRs, hand-over of the last chunk ----
    T = X if (n_groups - 1) % 2 == 0 else Y
    emit("s_nop 15")
    emit("s_nop 3")
    for c, name in enumerate(("cr", "cg", "cb")):
        emit(f"v_mov_b32 %[{name}], v{T + c}")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_barrier")
    for cc, p in dma_pieces(n_chunks + 1):
        dma_piece(cc, p)
    emit("s_mov_b32 m0, %[m0s]")
    return groups, chunks, out


def main():
    groups, chunks, lines = generate()
    if "bar" in ABLATE:
        lines = [l for l in lines if l != "s_barrier"]
    p = print
    p("// GENERATED by gen_fwd_stream_asm.py -- do not edit.  See that file for the design.")
    p("#pragma once\n")
    p("namespace nerf {\n")
    p(f"static_assert(plan::kFwdChunks.n_chunks == {len(chunks)} && plan::kFwdChunks.n_groups == {len(groups)}, \"stream plan\");")
    for i, c in enumerate(chunks):
        p(f"static_assert(plan::kFwdChunks.chunk_frag0[{i}] == {c['frag0']} && plan::kFwdChunks.chunk_count[{i}] == {c['count']}, \"stream plan\");")
    p(f"static_assert(plan::kChunkFrags == {CHUNK}, \"stream plan\");\n")
    p("// ab0/ab1: LDS byte address of ring slot 0/1 + lane*16;")
    p("// bb: LDS address of the bias table + 16*half; voff = wave*1024 + lane*16; ldsw = ring base +")
    p("// wave*1024 (wave-uniform); src = forward fragment stream.")
    clob = ", ".join(f'"v{i}"' for i in range(X, 256))
    p("__device__ __forceinline__ void fwd_stream_pass(unsigned ab0, unsigned ab1, unsigned bb, const bf16x8 (&x)[4],")
    p("                                                const bf16x8 (&d)[2], const char* src, unsigned voff, unsigned ldsw,")
    p("                                                float& sg, float& cr, float& cg, float& cb) {")
    p("  bf16x8 " + ", ".join(f"w{i}" for i in range(D)) + ";")
    p("  unsigned va, m0s;")
    p("  asm volatile(")
    for ln in lines:
        if ln.startswith(";"):
            p(f"      // {ln[2:]}")
        else:
            p(f'      "{ln}\\n\\t"')
    outs = [f'[w{i}] "=&v"(w{i})' for i in range(D)] + [
        '[va] "=&v"(va)', '[m0s] "=&s"(m0s)', '[sg] "=&v"(sg)', '[cr] "=&v"(cr)', '[cg] "=&v"(cg)', '[cb] "=&v"(cb)']
    ins = ['[ab0] "v"(ab0)', '[ab1] "v"(ab1)', '[bb] "v"(bb)'] + [f'[x{i}] "v"(x[{i}])' for i in range(4)] + [
        f'[d{i}] "v"(d[{i}])' for i in range(2)] + ['[src] "s"(src)', '[voff] "v"(voff)', '[ldsw] "s"(ldsw)']
    p("      : " + ", ".join(outs))
    p("      : " + ", ".join(ins))
    p(f'      : "memory", "scc", {clob});')
    p("}\n")
    p("}  // namespace nerf")
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    print(f"groups {len(groups)} chunks {[c['count'] for c in chunks]} lines {len(lines)} mfma {n_mfma}", file=sys.stderr)


if __name__ == "__main__":
    main()
