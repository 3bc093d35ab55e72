// svgf_atrous_lane_tfused.inc.h — EXPERIMENTS build only (-DSVGF_BUILD_EXPERIMENTS): the temporal pass fused into the lane kernel's
// staging path (FUSED = 1 / 2, DESIGN.md 5.8: built, parity-green, 176 us against 97 for the two kernels it replaces — parked).
// Textually included INSIDE k_atrous_lane (svgf_atrous_lane_impl.h), between the loader invariants and the prologue: it uses the
// kernel's locals (a, ta, gm, smem, slot_mod, rec_of, xs_of, stamp, ring_advance, ...) and, for TFUSED instantiations, runs the
// prologue and the loader loop itself (the loader waves return from inside it).
    // ====================================================================================================================
    // FUSED: the temporal pass (reference BackProjection, src/denoise.cu:185-317) runs inside this level's STAGING path.
    // A staged pixel is not loaded from a colour plane: the staging thread accumulates it — the arithmetic of k_temporal,
    // function for function (svgf_temporal.h) — from the frame's inputs and the history planes, writes the accumulated
    // colour + variance into the ring record (nowhere else: level 1's output is the next colour history, :391) and, for
    // the pixels the workgroup OWNS (its output rows and columns; halo pixels are recomputed by the neighbours), the
    // moments, the history length and the split G-buffer planes.  A pixel passes three stages, each one memory round trip:
    //   A  primary loads: normal / position / geomId (AoS texel or the producer's planes), history length;
    //   B  reprojection; loads of previous-frame geomId + normal of the four bilinear taps (isReprjValid :172-182);
    //   C  consistency test; then, in ONE batch for both pixels of a thread, the 1-spp colour and the history of the bilinear
    //      quad (:234-259) or the consistency data of the 3x3 fallback's other five taps (:262-286, whose history follows row
    //      by row, blocking); blend (:288-315); ring record + planes.
    // What a pixel carries from stage to stage is kept small on purpose — the loader waves have the 168 registers three waves
    // per SIMD leave: the colour is fetched in stage C rather than in A (3 registers), ring record and pixel index are re-derived.
    // ====================================================================================================================
    struct TAu_ { float nx, ny, nz, px, py, pz; int gid, N; };                        // stage A's data as the later stages use it
    // Carried values keep the SHAPE of the loads that fetch them (vector types, one per load instruction): a scalar array filled
    // from a wide load makes the register allocator load into a scratch tuple and copy — behind an s_waitcnt for a load it has
    // just issued (r04_exp_fused_v7: two exposed round trips per sub-step in stage B alone).  The pointer types say "4-byte
    // aligned": the plane elements are.
    typedef float v3f __attribute__((ext_vector_type(3)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef v2f __attribute__((aligned(4))) v2f_u;
    typedef v3f __attribute__((aligned(4))) v3f_u;
    typedef v4f __attribute__((aligned(4))) v4f_u;
    typedef v2i __attribute__((aligned(4))) v2i_u;
    // stage A as loaded: AoS texel = floats 0..3 {nx, ny, nz, px} + 4..5 {py, pz} + geomId; planes = normal + position + geomId
    struct TA_ { v4f g4; v2f g2; v3f n3, q3; int gid, N; };
    [[maybe_unused]] auto t_unpack = [](const TA_ &t) -> TAu_ {
        if constexpr (FUSED == 1) return TAu_{ t.g4.x, t.g4.y, t.g4.z, t.g4.w, t.g2.x, t.g2.y, t.gid, t.N };
        else return TAu_{ t.n3.x, t.n3.y, t.n3.z, t.q3.x, t.q3.y, t.q3.z, t.gid, t.N };
    };
    // stage B: the bilinear quad, one row piece per window row yy = 0, 1: geomId of taps (0, yy), (1, yy); their normals as 4 + 2 floats
    struct TB_ { v2i g[2]; v4f n4[2]; v2f n2[2]; };
    // Stage C1's request, ONE static shape of nine loads whatever the pixel needs (a load count that depended on the pixel would
    // make every later s_waitcnt conservative, see the pipeline below): five 12-byte, two 16-byte and two 8-byte pieces whose
    // ADDRESSES are per lane.  mode 1 (bilinear): colour, moments, history length of the quad's four taps.  mode 2 (3x3 fallback):
    // normal and geomId of the other five window taps 0, 1, 2, 3, 6.  mode 0: whatever lies at clamped addresses, never read.
    // (as vectors: h3[k] = colour of quad tap k | normal of extra tap k (k = 4: extra tap 6 only); h4[yy] = moments of quad taps
    // (0, yy), (1, yy) | h4[0].xyz = geomIds of window row -1; h2[yy] = history lengths of those two taps | h2[yy].x = geomId of
    // window tap 3 (yy = 0), 6 (yy = 1))
    struct TC_ { v3f h3[5]; v4f h4[2]; v2f h2[2]; v3f rgb; SvgfReproj rp; int mode, m9; };
    // The 3x3 window around the reprojected position (fx, fy): element index (+ 2) of the first column (fx - 1) of its rows
    // fy - 1, fy, fy + 1.  Rows are clamped into the image, the first column into [-2, W - 1]: a window with any tap on screen has
    // fx in [-1, W] and is read where it lies — its off-screen taps fall at most two elements outside a row, i.e. inside the
    // planes' padding at the two ends of a plane (svgf_api.hip: kPlanePad) — and every tap's validity comes from t_on_screen()
    // on the same coordinates, never from the address.  The + 2 makes it a non-negative offset from (plane - 2 elements).
    // Integer arithmetic on (fx, fy) clamped to [-2, W + 1] x [-2, H + 1] (NaN -> -2): exact for every window that has a tap on
    // screen, and all the others only need SOME address inside the plane.
    struct TWin_ { int ix, iy; unsigned em, e0, e1; };
    [[maybe_unused]] auto t_window = [&](float fx, float fy) -> TWin_ {
        TWin_ w;
        w.ix = (int)fminf(fmaxf(fx, -2.0f), (float)(W + 1));
        w.iy = (int)fminf(fmaxf(fy, -2.0f), (float)(H + 1));
        const int col = min(max(w.ix, -1), W) + 1;
        w.em = (unsigned)(__mul24(min(max(w.iy - 1, 0), H - 1), W) + col);      // (W * H < 2^24: atrous_fused_supported)
        w.e0 = (unsigned)(__mul24(min(max(w.iy, 0), H - 1), W) + col);
        w.e1 = (unsigned)(__mul24(min(max(w.iy + 1, 0), H - 1), W) + col);
        return w;
    };
    // Taps of the window on screen (the bounds part of isReprjValid, :173-176; svgf_tap_index() >= 0 for the nine taps), bit
    // (yy + 1) * 3 + (xx + 1).  Separable: bit k of span3(i, n) says 0 <= i + k - 1 < n.  With i clamped to [-2, n + 1] as above
    // the answer is the one the float comparisons of svgf_tap_index() give: inside the clamp range fx + xx is exact, outside it
    // (and for NaN, which the clamp sends to -2) no tap of that axis is on screen.
    [[maybe_unused]] auto t_on_screen = [&](const TWin_ &w) -> int {
        auto span3 = [](int i, int n) { return (7 << min(max(1 - i, 0), 3)) & (7 >> (2 - min(n - i, 2))) & 7; };
        const int mx = span3(w.ix, W), my = span3(w.iy, H);
        return (mx * 0x49) & (((my & 1) | ((my & 2) << 2) | ((my & 4) << 4)) * 7);
    };
    // history planes are addressed as (the context's allocation) + 32-bit byte offset, see LaneFused
    [[maybe_unused]] auto t_ptr = [&](unsigned byte_off) -> const char * {
        if constexpr (TFUSED) return reinterpret_cast<const char *>(ta.arena) + byte_off;
        else return nullptr;
    };
    [[maybe_unused]] auto t_wptr = [&](unsigned byte_off) -> char * {
        if constexpr (TFUSED) return const_cast<char *>(reinterpret_cast<const char *>(ta.arena)) + byte_off;
        else return nullptr;
    };
    constexpr int kLdsDump = RING_BYTES + 32;          // 48 bytes nobody reads: the ring record of a thread's idle pixel slot
    // pixel (lattice row br, y-phase yp, staged column xi) -> ring record, pixel index, flags (1: inside the image, 2: owned)
    [[maybe_unused]] auto t_describe = [&](int br, int yp, int xi, bool active, int &lds_off, unsigned &p, int &flags) {
        const int y = phase + yp + (br << LOG2S);
        const int xs = xs_of(xi);
        const bool ok = active && (br >= 0) && (y < H) && (xs >= 0) && (xs < W);
        const bool owned = ok && (br >= b0) && (br < b1) && (xi >= 2 * S) && (xi < 2 * S + TXW);
        lds_off = active ? (slot_mod(br) * YP + yp) * ROWB + rec_of(xi) : kLdsDump;
        p = (unsigned)min(max(y, 0), H - 1) * (unsigned)W + (unsigned)min(max(xs, 0), W - 1);
        flags = (ok ? 1 : 0) | (owned ? 2 : 0);
    };
    [[maybe_unused]] auto t_stage_a = [&](TA_ &t, unsigned p) __attribute__((always_inline)) {
        if constexpr (TFUSED) {
            // (unsigned 32-bit byte offsets: scalar base + vector offset addressing; W * H * 52 < 2^32 is checked by the launcher)
            if constexpr (FUSED == 1) {         // the boundary's AoS texels
                const char *g = reinterpret_cast<const char *>(ta.gbuf) + __umul24(p, 52u);
                t.g4 = *reinterpret_cast<const v4f_u *>(g);
                t.g2 = *reinterpret_cast<const v2f_u *>(g + 16);
                t.gid = *reinterpret_cast<const int *>(g + 48);
            } else {                            // planes written in place by the producer (svgf_planar_gbuffer)
                t.n3 = *reinterpret_cast<const v3f_u *>(reinterpret_cast<const char *>(ta.nrm_cur) + __umul24(p, 12u));
                t.q3 = *reinterpret_cast<const v3f_u *>(reinterpret_cast<const char *>(ta.pos_cur) + __umul24(p, 12u));
                t.gid = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(ta.gid_cur) + p * 4u);
            }
            t.N = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(ta.hlen) + p * 4u);
        }
    };
    // history lookup wanted (:197): the pixel is inside the image, has a history and hit something
    [[maybe_unused]] auto t_wants_history = [&](const TAu_ &a_, int flags) { return (flags & 1) && a_.N > 0 && a_.gid != -1; };
    // (unconditional: a pixel that wants no history reprojects whatever its position is — NaN included — and reads clamped
    // addresses; nothing of it is used)
    [[maybe_unused]] auto t_stage_b = [&](const TA_ &araw, TB_ &b_, SvgfReproj &rp_out) __attribute__((always_inline)) {
        if constexpr (TFUSED) {
            const TAu_ a_ = t_unpack(araw);
            const SvgfReproj rp = svgf_reproject(ta, a_.px, a_.py, a_.pz);
            rp_out = rp;
            const TWin_ w = t_window(rp.fx, rp.fy);
#pragma unroll
            for (int yy = 0; yy <= 1; yy++) {                             // the bilinear quad: taps (0, yy), (1, yy), one row piece each
                const unsigned e = (yy ? w.e1 : w.e0) + 1u;
                const char *np = t_ptr(ta.o_nrm_prev + __umul24(e, 12u));
                b_.g[yy] = *reinterpret_cast<const v2i_u *>(t_ptr(ta.o_gid_prev + e * 4u));
                b_.n4[yy] = *reinterpret_cast<const v4f_u *>(np);
                b_.n2[yy] = *reinterpret_cast<const v2f_u *>(np + 16);
            }
        }
    };
    [[maybe_unused]] auto t_stage_c1 = [&](const TA_ &araw, const TB_ &b_, const SvgfReproj &rp, int flags, unsigned p, TC_ &c_) __attribute__((always_inline)) {
        if constexpr (TFUSED) {
            const TAu_ a_ = t_unpack(araw);
            c_.rgb = *reinterpret_cast<const v3f_u *>(reinterpret_cast<const char *>(ta.in_rgb) + __umul24(p, 12u));
            c_.rp = rp;
            const bool wants = t_wants_history(a_, flags);
            const TWin_ w = t_window(rp.fx, rp.fy);
            int m9 = wants ? t_on_screen(w) : 0;                          // taps on screen (:173-176), bit (yy + 1) * 3 + xx + 1
            // the bilinear quad: window taps 4, 5, 7, 8; its geomIds and normals are here: consistency test (:177-180)
            constexpr int quad[4] = { 4, 5, 7, 8 };
            bool all = wants;                                             // (tap 4 on screen is fx, fy inside the image, :230)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int yy = k >> 1;
                const bool right = (k & 1) != 0;      // tap (1, yy): the second geomId, the floats 3, 4, 5 of the row piece
                const bool ok = ((m9 >> quad[k]) & 1) & svgf_tap_consistent(right ? b_.g[yy].y : b_.g[yy].x, right ? b_.n4[yy].w : b_.n4[yy].x,
                                                                            right ? b_.n2[yy].x : b_.n4[yy].y, right ? b_.n2[yy].y : b_.n4[yy].z, a_.gid, a_.nx, a_.ny, a_.nz);
                m9 = ok ? m9 : (m9 & ~(1 << quad[k]));
                all = all & ok;
            }
            c_.m9 = m9;
            c_.mode = wants ? (all ? 1 : 2) : 0;
            const unsigned em = w.em, e0 = w.e0, e1 = w.e1;
            // five 12-byte pieces: the quad's history colours, or the normals of window taps 0, 1, 2, 3, 6
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const unsigned hist = ta.o_cv_hist + ((k < 2 ? e0 : e1) + 1u + (unsigned)(k & 1)) * 16u;
                const unsigned nrmp = ta.o_nrm_prev + __umul24(k < 3 ? em + (unsigned)k : (k == 3 ? e0 : e1), 12u);
                c_.h3[k] = *reinterpret_cast<const v3f_u *>(t_ptr((all && k < 4) ? hist : nrmp));
            }
            // two 16-byte pieces: the quad's history moments (two taps a row), or the geomIds of window row -1
#pragma unroll
            for (int yy = 0; yy <= 1; yy++)
                c_.h4[yy] = *reinterpret_cast<const v4f_u *>(t_ptr(all ? ta.o_mom_hist + ((yy ? e1 : e0) + 1u) * 8u : ta.o_gid_prev + em * 4u));
            // two 8-byte pieces: the quad's history lengths, or the geomIds of window taps 3 and 6 (first element)
#pragma unroll
            for (int yy = 0; yy <= 1; yy++)
                c_.h2[yy] = *reinterpret_cast<const v2f_u *>(t_ptr((all ? ta.o_hlen + 4u : ta.o_gid_prev) + (yy ? e1 : e0) * 4u));
        }
    };
    [[maybe_unused]] auto t_stage_c2 = [&](const TA_ &araw, const TC_ &c_, int flags, unsigned p, int lds_off) __attribute__((always_inline)) {
        if constexpr (TFUSED) {
            const TAu_ a_ = t_unpack(araw);
            const float lum = svgf_lum_strict(c_.rgb.x, c_.rgb.y, c_.rgb.z);
            SvgfHistSum hs = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
            bool valid = false, do_div = false;       // one division of the six sums, whichever path filled them
            float dsum = 0.0f;
            if (c_.mode == 1) {                                       // bilinear (:234-259)
                float w[4];
                svgf_bilinear_weights(c_.rp.fracx, c_.rp.fracy, w);
                float sumw = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int yy = k >> 1;
                    const bool right = (k & 1) != 0;
                    svgf_hist_add_weighted(hs, w[k], c_.h3[k].x, c_.h3[k].y, c_.h3[k].z, right ? c_.h4[yy].z : c_.h4[yy].x, right ? c_.h4[yy].w : c_.h4[yy].y,
                                           __float_as_int(right ? c_.h2[yy].y : c_.h2[yy].x));
                    sumw += w[k];
                }
                dsum = sumw; do_div = (double)sumw >= 0.01;
                valid = true;
            } else if (c_.mode == 2) {                                // 3x3 box around floor (:262-286)
                int m9 = c_.m9;
                constexpr int extra[5] = { 0, 1, 2, 3, 6 };
                const int ge[5] = { __float_as_int(c_.h4[0].x), __float_as_int(c_.h4[0].y), __float_as_int(c_.h4[0].z), __float_as_int(c_.h2[0].x), __float_as_int(c_.h2[1].x) };
#pragma unroll
                for (int j = 0; j < 5; j++)
                    if (!svgf_tap_consistent(ge[j], c_.h3[j].x, c_.h3[j].y, c_.h3[j].z, a_.gid, a_.nx, a_.ny, a_.nz)) m9 &= ~(1 << extra[j]);
                if (m9) {                                             // a consistent tap exists: the window's history, one row piece at a time
                    // (The one place of the pipeline that waits for loads it has just issued: rare in steady state — a pixel next to
                    // an edge whose reprojection still finds part of its surface.  The branch ends with nothing of it in flight, so
                    // the s_waitcnt bookkeeping of the main path is the same whether it was taken or not.)
                    float cnt = 0.0f;
#pragma unroll 1
                    for (int yy = -1; yy <= 1; yy++) {
                        const int mrow = (m9 >> ((yy + 1) * 3)) & 7;
                        if (mrow) {
                            const TWin_ w = t_window(c_.rp.fx, c_.rp.fy);
                            const unsigned e = yy < 0 ? w.em : (yy == 0 ? w.e0 : w.e1);
                            const float *ch = reinterpret_cast<const float *>(t_ptr(ta.o_cv_hist + e * 16u));
                            const float *m = reinterpret_cast<const float *>(t_ptr(ta.o_mom_hist + e * 8u));
                            const int *l = reinterpret_cast<const int *>(t_ptr(ta.o_hlen + e * 4u));
                            float c3[3][3], mo3[3][2];
                            int l3[3];
#pragma unroll
                            for (int xx = 0; xx < 3; xx++) {
                                c3[xx][0] = ch[4 * xx]; c3[xx][1] = ch[4 * xx + 1]; c3[xx][2] = ch[4 * xx + 2];
                                mo3[xx][0] = m[2 * xx]; mo3[xx][1] = m[2 * xx + 1];
                                l3[xx] = l[xx];
                            }
#pragma unroll
                            for (int xx = 0; xx < 3; xx++)            // raster order, as the reference sums
                                if ((mrow >> xx) & 1) { svgf_hist_add(hs, c3[xx][0], c3[xx][1], c3[xx][2], mo3[xx][0], mo3[xx][1], l3[xx]); cnt += 1.0f; }
                        }
                    }
                    dsum = cnt; do_div = true;
                    valid = true;
                }
            }
            if (do_div) svgf_hist_div(hs, dsum);
            const SvgfTemporalOut o = svgf_temporal_blend(ta, c_.rgb.x, c_.rgb.y, c_.rgb.z, lum, a_.N, valid, hs);
            // ring record, as rows_store() writes it (an idle slot's goes to kLdsDump)
            const bool ok = (flags & 1) != 0;
            const float inf = __builtin_huge_valf();
            const float mag = fabsf(a_.nx) + fabsf(a_.ny) + fabsf(a_.nz) + fabsf(a_.px) + fabsf(a_.py) + fabsf(a_.pz);
            if (ok && !(mag < inf)) *nan_seen = 1;
            char *d = smem + lds_off;
            *reinterpret_cast<float4 *>(d) = make_float4(a_.nx, a_.px, a_.ny, a_.py);
            *reinterpret_cast<float4 *>(d + 16) = make_float4(a_.nz, a_.pz, ok ? lum_f64(o.cv.x, o.cv.y, o.cv.z) : inf, 0.0f);
            *reinterpret_cast<float4 *>(d + 32) = ok ? o.cv : make_float4(0.f, 0.f, 0.f, 0.f);
            // The planes the rest of the frame (and the next) reads: written for the pixels this workgroup OWNS.  Every thread
            // issues every store — a pixel that is not owned stores into a scrap buffer (ta.dump) — so that the NUMBER of
            // vector-memory operations of a sub-step does not depend on the pixel (see the pipeline below).
            const bool owned = (flags & 2) != 0;
            const unsigned scrap = ta.o_dump + (unsigned)(tid & 255) * 16u;
            *reinterpret_cast<int *>(t_wptr(owned ? ta.o_hlen_upd + p * 4u : scrap)) = o.hlen;
            *reinterpret_cast<float2 *>(t_wptr(owned ? ta.o_mom_acc + p * 8u : scrap)) = o.mom;
            // (the accumulated plane itself: only when something besides this level reads it — else o_cv_acc is the scrap buffer)
            *reinterpret_cast<float4 *>(t_wptr(owned ? ta.o_cv_acc + (ta.cv_acc ? p * 16u : (unsigned)(tid & 255) * 16u) : scrap)) = o.cv;
            if constexpr (FUSED == 1) {
                float *n = reinterpret_cast<float *>(t_wptr(owned ? ta.o_nrm_cur + __umul24(p, 12u) : scrap));
                float *q = reinterpret_cast<float *>(t_wptr(owned ? ta.o_pos_cur + __umul24(p, 12u) : scrap));
                n[0] = a_.nx; n[1] = a_.ny; n[2] = a_.nz; q[0] = a_.px; q[1] = a_.py; q[2] = a_.pz;
                *reinterpret_cast<int *>(t_wptr(owned ? ta.o_gid_cur + p * 4u : scrap)) = a_.gid;
            }
        }
    };

    if constexpr (TFUSED) {
        // One software pipeline for the prologue and the loop, ONE pixel per thread and sub-step.  Pixel q of a thread:
        //   q = 0 .. 3     the ten prologue rows b0-2 .. b0+2 (both y-phases), dealt over all 768 threads, four pixels each;
        //   q >= 4         lattice row b0+3 + (q-4)/2, y-phase (q-4) & 1, at the staged column a LOADER thread owns.
        // Sub-step u runs stage C2 of pixel u-4, C1 of pixel u-3, B of pixel u-2 and A of pixel u: every stage's loads have at
        // least one sub-step to land.  After sub-step 7 the prologue rows are in the ring (first barrier; the compute threads
        // leave for their warm-up rows); sub-steps 8+2i, 9+2i are iteration i of the loop: they commit the two y-phases of
        // lattice row bo+3 while the compute waves work on output row bo = b0+i.
        // What makes or breaks it is s_waitcnt: vmcnt counts loads AND stores, in order, and the compiler can only wait for
        // "all but the N youngest" when it knows N.  A vector-memory operation under a condition — a store for owned pixels
        // only, history loads for pixels that have a history — makes N unknown, every wait becomes vmcnt(0), and vmcnt(0)
        // behind freshly issued stores waits for their acknowledgements: 3-5 thousand ticks, twice a sub-step
        // (profiles/r04_exp_fused_v5_timeline.log, r04_exp_fused_v6_timeline.log: 213-219 us per 1080p frame).  So every
        // sub-step of the loop issues the SAME operations whatever its pixel is: loads from clamped addresses, stores into a
        // scrap buffer, one request shape for stage C1; the buffers of stage A alternate statically (loop unrolled by two:
        // rotating them by assignment copies registers whose loads are in flight); one pixel per stage keeps the carried
        // state at ~85 registers (two pixels per stage did not fit the 168 that three waves per SIMD leave).
        const int lt = tid - NC;                        // loader thread index (loader threads only)
        const int lxi = min(max(lt, 0), RW - 1);        // the staged column a loader thread owns in the loop
        const bool lactive = is_loader && lt < RW;
        auto describe_px = [&](int q, int &lo, unsigned &p, int &fl) __attribute__((always_inline)) {
            int br, yp, xi;
            bool active;
            if (q < 4) {
                const int idx = tid + q * NT;
                active = idx < 5 * YP * RW;
                const int idc = min(idx, 5 * YP * RW - 1);
                const int rr = idc / RW;
                xi = idc - rr * RW;
                br = b0 - 2 + rr / YP; yp = rr % YP;
            } else {
                br = b0 + 3 + ((q - 4) >> 1); yp = (q - 4) & 1; xi = lxi;
                active = lactive && br <= b1 + 1;
            }
            t_describe(br, yp, xi, active, lo, p, fl);
        };
        // The loop's pixels (q >= 4) differ only by their row: what depends on the thread is computed once.  l_flags: bit 0 the
        // column is inside the image, bit 1 it is one of the workgroup's output columns (0 for a thread without a column).
        const int l_xs = xs_of(lxi);
        const unsigned l_xsc = (unsigned)min(max(l_xs, 0), W - 1);
        const int l_rec = rec_of(lxi);
        const int l_flags = (lactive && l_xs >= 0 && l_xs < W) ? (1 | ((lxi >= 2 * S && lxi < 2 * S + TXW) ? 2 : 0)) : 0;
        auto describe_loop_px = [&](int q, int &lo, unsigned &p, int &fl) __attribute__((always_inline)) {
            const int br = b0 + 3 + ((q - 4) >> 1), yp = q & 1;                 // wave-uniform, like everything derived from them
            const int y = phase + yp + (br << LOG2S);
            const bool row_live = br <= b1 + 1;
            fl = l_flags & ((row_live && y < H) ? (br < b1 ? 3 : 1) : 0);
            p = (unsigned)(min(y, H - 1) * W) + l_xsc;
            lo = (row_live && lactive) ? (slot_mod(br) * YP + yp) * ROWB + l_rec : kLdsDump;
        };
        TA_ a_even, a_odd;                              // stage A of the pixels with even / odd index
        TA_ ab; TB_ bb; SvgfReproj rb;                  // the pixel between stages B and C1
        TA_ ac; TC_ cc;                                 // the pixel between stages C1 and C2
        if (is_loader) __builtin_amdgcn_s_setprio(SVGF_LANE_LOADER_PRIO);
        const int n_sub = 8 + 2 * (b1 - b0);
        // abuf: holds pixel u-2 on entry (stage B reads it), pixel u on exit.  PRO: the prologue's sub-steps 0 .. 7, where the
        // pipeline fills and all 768 threads take part; else the loop's, loader threads only, every stage every time.
        auto substep = [&](int u, TA_ &abuf, auto pro_tag) __attribute__((always_inline)) {
            constexpr bool PRO = decltype(pro_tag)::value;
            if (!PRO && !(u & 1)) stamp(0);
            if (!PRO || u >= 4) {                               // C2: blend, ring record + planes
                int lo, fl; unsigned p;
                if constexpr (PRO) describe_px(u - 4, lo, p, fl); else describe_loop_px(u - 4, lo, p, fl);
                t_stage_c2(ac, cc, fl, p, lo);
            }
            if (!PRO && !(u & 1)) stamp(1);
            if (!PRO || (u >= 3 && (u - 3 < 4 || is_loader))) { // C1: consistency, history request
                int lo, fl; unsigned p;
                if constexpr (PRO) describe_px(u - 3, lo, p, fl); else describe_loop_px(u - 3, lo, p, fl);
                ac = ab;
                t_stage_c1(ac, bb, rb, fl, p, cc);
            }
            if (!PRO && !(u & 1)) stamp(2);
            if (!PRO || (u >= 2 && (u - 2 < 4 || is_loader))) { // B: reprojection, consistency data of the bilinear quad
                ab = abuf;
                t_stage_b(ab, bb, rb);
            }
            if (!PRO && !(u & 1)) stamp(3);
            if (!PRO || u < 4 || is_loader) {                   // A: primary loads
                int lo, fl; unsigned p;
                if constexpr (PRO) describe_px(u, lo, p, fl); else describe_loop_px(u, lo, p, fl);
                t_stage_a(abuf, p);
            }
            if (!PRO && !(u & 1)) stamp(4);
        };
#pragma unroll 1
        for (int u = 0; u < 8; u += 2) {
            substep(u, a_even, std::true_type{});
            substep(u + 1, a_odd, std::true_type{});
        }
        __syncthreads();
        stamp_at(1);
        if (is_loader) {
            // The loop's first iteration is peeled: the s_waitcnt pass merges the states of a loop header's predecessors, and a wait
            // for "all but the N youngest" survives the merge only if N is the same on both — with the prologue's branchy state
            // (or an empty one) on one side, the first waits of every iteration came out as vmcnt(0): one drained pipeline per
            // iteration.  Entered from a copy of its own body, the header sees the same sequence on both edges.
            auto iteration = [&](int u) __attribute__((always_inline)) {
                substep(u, a_even, std::false_type{});
                substep(u + 1, a_odd, std::false_type{});
                stamp(5);
                __syncthreads();
                stamp(6);
                ring_advance(); dbg_it++;
            };
            iteration(8);
#pragma unroll 1
            for (int u = 10; u < n_sub; u += 2) iteration(u);
            return;
        }
    }
