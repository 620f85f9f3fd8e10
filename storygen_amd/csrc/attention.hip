// Fused (flash-style) attention forward for gfx950: O = softmax(scale * Q K^T) V, fp16 in/out, fp32 softmax.
//
// Serves the three attentions of StoryGen's BasicTransformerBlock (model/attention.py:255-260 self, :271-276 text,
// :285-290 image / Visual-Language Context) — head dim D = C/8 in {40, 80, 160}, Nk = HW, 77 or R*HW.
//
// Operands: Q and K token-major ([B, N, H*D], heads interleaved in the channel dimension exactly as
// head_to_batch_dim expects) and V *transposed*: VT[b][h*D + d][key].  The V projection is a GEMM anyway, so the host
// simply computes it with swapped operands (VT = Wv . X^T) — the attention kernel then needs no in-kernel transpose
// and every LDS tile it consumes is a plain 16-byte-chunk copy of global memory, which is what LDS-DMA wants.
//
// Work decomposition: one workgroup = NW wave64, each wave owns 32 queries of one (batch, head); K / VT are streamed in
// tiles of 64 keys through an S-stage LDS ring filled by LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging),
// counted vmcnt waits and ONE raw s_barrier per tile (same pipeline as the GEMM mainloop).
//
// MFMA formulation (v_mfma_f32_32x32x16_f16), chosen so that softmax never leaves registers:
//   S^T[key, q] = sum_d K[key, d] Q[q, d]      A = K fragment (LDS, ds_read_b128), B = Q^T fragment (registers)
//     The A rows are fed in a permuted order (row i of the MFMA reads key pi(i), pi = swap of bits 2 and 3), so that
//     accumulator register r of lane (q = l & 31, hi = l >> 5) holds key 16 (r >> 3) + 8 hi + (r & 7): the 8 registers
//     of a 16-key step are 8 CONSECUTIVE keys.  Row max / row sum are in-lane reductions plus ONE exchange with l^32.
//   O^T[d, q] = sum_key VT[d, key] P^T[key, q]  A = VT fragment (LDS, one ds_read_b128 of 8 consecutive keys),
//     B = P^T = the lane's own accumulator registers converted to fp16.  P never moves across lanes or through LDS.
//   Head dim 40 is zero-padded to 48 on the contraction side (the Q fragment of chunk 5 is zero; the K fragment reads
//   the first 16 bytes of the next LDS row, finite data) and to 64 on the O^T rows (rows >= D are never stored).
// LDS images (both filled by LDS-DMA, so linear in lane order; the swizzles are applied on the SOURCE address and again
// on the read, guide §5.4 rule 21):
//   K tile  [64 keys][D halves], row stride 2D bytes, chunk c of row k at slot c ^ kswz(k): conflict-free ds_read_b128
//           (kswz = 0 for D=40 whose 80-byte stride already spreads 16 rows over 16 slots, (k>>3)&1 for D=80,
//           (k>>2)&3 for D=160).
//   VT tile [D rows][64 keys], row stride 128 bytes, chunk c of row d at slot c ^ ((d>>1)&7).
// Online softmax runs in the log2 domain with the scale folded into the exponent's FMA, and rescales the accumulators
// only when some row's running max grows by more than 2^6 (wave-uniform branch; P <= 64 stays exact enough in fp16
// and the row sums are fp32).
#include "attention_kernel.h"

using namespace sgattn;


// Validates a descriptor and translates it into kernel parameters.
static int attn_params(const sg_attn_desc* d, AttnParams& p, const char* who) {
    SG_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SG_REQUIRE(d->q && d->k && d->vt && d->o, "%s: null q/k/vt/o", who);
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, "%s: bad shape", who);
    if (d->D != 40 && d->D != 80 && d->D != 160)
        return sg_set_error(SG_EUNSUP, "%s: head dim %d not in {40, 80, 160}", who, d->D);
    SG_REQUIRE(d->kv_batches >= 0 && d->kv_batches <= d->B, "%s: kv_batches must be in [0, B]", who);
    SG_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 8 == 0 && d->ldo % 4 == 0, "%s: row strides", who);
    SG_REQUIRE(d->bsq % 8 == 0 && d->bsk % 8 == 0 && d->bsvt % 8 == 0 && d->bso % 4 == 0, "%s: batch strides", who);
    SG_REQUIRE(sg_aligned16(d->q) && sg_aligned16(d->k) && sg_aligned16(d->vt) && sg_aligned16(d->o), "%s: 16-byte alignment", who);
    const int64_t hd = (int64_t)d->H * d->D;
    SG_REQUIRE(d->ldq >= hd && d->ldk >= hd && d->ldo >= hd, "%s: token stride smaller than H*D", who);
    SG_REQUIRE(d->ldvt >= ((d->Nk + 7) & ~7), "%s: ldvt must cover Nk rounded up to 8 keys", who);
    SG_REQUIRE((int64_t)d->D * d->ldvt < (1ll << 31), "%s: VT head slab too large for 32-bit offsets", who);
    p = AttnParams{};
    p.q = reinterpret_cast<const f16*>(d->q); p.ldq = d->ldq; p.bsq = d->bsq;
    p.k = reinterpret_cast<const f16*>(d->k); p.ldk = d->ldk; p.bsk = d->bsk;
    p.vt = reinterpret_cast<const f16*>(d->vt); p.ldvt = d->ldvt; p.bsvt = d->bsvt;
    p.o = reinterpret_cast<f16*>(d->o); p.ldo = d->ldo; p.bso = d->bso;
    p.B = d->B; p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk;
    p.kv_batches = d->kv_batches > 0 ? d->kv_batches : d->B;
    p.scale_log2 = d->scale * 1.44269504088896340736f;
    if (d->kv2_batches > 0) {      // leading K/V rows with their own key count (same token / row strides)
        SG_REQUIRE(d->k2 && d->vt2 && d->Nk2 > 0 && d->kv2_batches <= p.kv_batches, "%s: kv2_batches needs k2, vt2, Nk2 and at most kv_batches rows", who);
        SG_REQUIRE(d->bsk2 % 8 == 0 && d->bsvt2 % 8 == 0 && sg_aligned16(d->k2) && sg_aligned16(d->vt2), "%s: k2 / vt2 strides and alignment", who);
        SG_REQUIRE(d->ldvt >= ((d->Nk2 + 7) & ~7), "%s: ldvt must cover Nk2 rounded up to 8 keys", who);
        p.k2 = reinterpret_cast<const f16*>(d->k2); p.bsk2 = d->bsk2;
        p.vt2 = reinterpret_cast<const f16*>(d->vt2); p.bsvt2 = d->bsvt2;
        p.Nk2 = d->Nk2; p.kv2 = d->kv2_batches;
    }
    return SG_OK;
}

// 4-wave workgroups with a 3-deep ring when that still gives the chip >= 2 workgroups per CU (a property of the queries only)
static bool attn_big(const sg_attn_desc* d) { return (long)sg_cdiv(d->Nq, 128) * d->H * d->B >= 512; }

static int attn_fwd(const sg_attn_desc* d, float* lse2, sg_stream_t stream) {
    AttnParams p;
    if (int rc = attn_params(d, p, "sg_attn_fwd_f16")) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (lse2) {   // training forward (sg_attn_fwd_lse_f16): the default instantiations with the log-sum-exp rows stored
        p.lse2 = lse2;
        if (d->D == 40) launch_attn<40, 4, 3, 1, false, true>(p, st);
        else if (d->D == 80) launch_attn<80, 4, 3, 1, false, true>(p, st);
        else launch_attn<160, 4, 3, 1, false, true>(p, st);
        SG_CHECK_LAUNCH("sg_attn_fwd_lse_f16");
        return SG_OK;
    }
    const bool big = attn_big(d);
    const SgOptions& opt = sg_options();          // development options (sg_debug_set_option), defaults in common.h
    const int sub2 = opt.attn_sub2, prio = opt.attn_prio;
    if (d->D == 40) {
        if (big && sub2 && d->Nk >= 256) launch_attn<40, 4, 2, 2>(p, st);   // 128 keys per barrier, 2-stage ring
        else if (big && prio) launch_attn<40, 4, 3, 1, true>(p, st);
        else if (big && opt.attn_d40_general) launch_attn<40, 4, 3, 1, false, false, false, true>(p, st);   // round-3 softmax (A/B)
        else if (big && opt.attn_lean) launch_attn<40, 4, 3, 1, false, false, true>(p, st);   // V^T fragments per k-step: fewer VGPRs
        else if (big) launch_attn<40, 4, 3>(p, st);
        else launch_attn<40, 2, 2>(p, st);
    }
    else if (d->D == 80) {
        // measured (tools/bench_norm.py --attn, B3 Nq1024 Nk3072): 2 waves x 2 stages 76.6 us, 2 x 3 59.8, 4 x 3 55.2 — the
        // deeper ring matters more than the number of workgroups here.  option attn_d80 (sg_debug_set_option): 1 = always 4 x 3
        // (default), 0 = 4 x 3 only when the grid fills the chip else 2 x 2, 2 = always 2 x 3
        const int v80 = opt.attn_d80;
        if (v80 == 1 || (v80 == 0 && big)) launch_attn<80, 4, 3>(p, st);
        else if (v80 == 2) launch_attn<80, 2, 3>(p, st);
        else launch_attn<80, 2, 2>(p, st);
    }
    else {
        // the 16x16 level has few workgroups with long key loops; measured (tools/bench_norm.py --attn, B3 Nq256 Nk768):
        // 2 waves x 2 stages 30.3 us, 2 x 3 30.2, 4 x 2 25.6, 4 x 3 25.4 -> 4 waves x 3 stages.  option attn_d160 = 0..3 picks
        // one of the four (development knob).
        // round 5: at Nq <= 256 (the 16x16 / 8x8 levels) the four waves of a workgroup split the KEYS of one 32-query block instead of
        // taking 32 queries each (attn_fwd_ksplit_kernel) — option attn_d160 = 4 (default); 0..3 = the query-split instantiations
        const int v160 = opt.attn_d160;
        // (only where it wins, tools/calls/r5_call09.sh: one round of workgroups — a 160 KB workgroup owns its CU — and at least two
        // tiles of keys; batch 20 of the batched reference pass, 1 280 workgroups, measured 42.7 vs 27.4 us)
        if (v160 == 4 && d->Nq <= 256 && d->Nk > KVBLK && (long)sg_cdiv(d->Nq, 32) * d->H * d->B <= 256) launch_attn_ksplit<160, 4>(p, st);
        else if (v160 == 1) launch_attn<160, 2, 3>(p, st);
        else if (v160 == 2) launch_attn<160, 4, 2>(p, st);
        else if (v160 == 3 || v160 == 4) launch_attn<160, 4, 3>(p, st);
        else launch_attn<160, 2, 2>(p, st);
    }
    SG_CHECK_LAUNCH("sg_attn_fwd_f16");
    return SG_OK;
}

extern "C" int sg_attn_fwd_f16(const sg_attn_desc* d, sg_stream_t stream) { return attn_fwd(d, nullptr, stream); }

// Two attentions over the same query geometry (B, H, Nq, D) in one launch — the text and the image cross-attention of one
// BasicTransformerBlock.  The longer key loop is numbered first.  Pairs that the default instantiation table would not serve with
// one kernel (development options set, different query geometry) are simply launched one after the other.
extern "C" int sg_attn_fwd_pair_f16(const sg_attn_desc* d0, const sg_attn_desc* d1, sg_stream_t stream) {
    AttnParams p0, p1;
    if (int rc = attn_params(d0, p0, "sg_attn_fwd_pair_f16[0]")) return rc;
    if (int rc = attn_params(d1, p1, "sg_attn_fwd_pair_f16[1]")) return rc;
    const SgOptions& opt = sg_options();
    const bool same = d0->D == d1->D && d0->B == d1->B && d0->H == d1->H && d0->Nq == d1->Nq;   // (short K/V rows: either problem)
    const bool defaults = !opt.attn_sub2 && !opt.attn_prio && opt.attn_d80 == 1 && opt.attn_d160 >= 3;
    if (!same || !defaults) {
        if (int rc = attn_fwd(d0, nullptr, stream)) return rc;
        return attn_fwd(d1, nullptr, stream);
    }
    const AttnParams& a = p0.Nk >= p1.Nk ? p0 : p1;
    const AttnParams& b = p0.Nk >= p1.Nk ? p1 : p0;
    hipStream_t st = (hipStream_t)stream;
    if (d0->D == 40) {
        if (attn_big(d0)) launch_attn_pair<40, 4, 3>(a, b, st);
        else launch_attn_pair<40, 2, 2>(a, b, st);
    } else if (d0->D == 80) launch_attn_pair<80, 4, 3>(a, b, st);
    else launch_attn_pair<160, 4, 3>(a, b, st);
    SG_CHECK_LAUNCH("sg_attn_fwd_pair_f16");
    return SG_OK;
}

extern "C" int sg_attn_fwd_lse_f16(const sg_attn_desc* d, float* lse2, sg_stream_t stream) {
    SG_REQUIRE(lse2 != nullptr, "sg_attn_fwd_lse_f16: null lse2");
    return attn_fwd(d, lse2, stream);
}
