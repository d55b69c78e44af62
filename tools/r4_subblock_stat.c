/* How many (entry, 8x8 block) visits of the blend kernels would a per-4x4-sub-block liveness test save?  (CPU estimate on oracle data.)
 * Forward: a wave visits entry e if e can reach its 8x8 block (alpha >= 1/255 somewhere in it) and the block is not all done;
 * with sub-blocks: only if it reaches a 4x4 sub-block that still has a live pixel.  Backward (back to front over k < hi): a lane is
 * active iff k < n_contrib; now: the quadrant has an active lane; sub-block: a reached sub-block has an active lane. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static float expf_(float x) { return expf(x); }
void subblock_stats(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D, const float* conic_opacity,
                    long long* out /* fwd: visits, sub4 visits, blending visits; bwd: visits, sub4 visits, blending visits; sub 8x4, 4x8 fwd */)
{
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    long long fv = 0, fs = 0, fb = 0, bv = 0, bs = 0, bb = 0, f84 = 0, f48 = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : fv, fs, fb, bv, bs, bb, f84, f48)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const uint32_t n = r1 - r0;
        if (!n) continue;
        float T[256]; uint8_t done[256]; uint32_t ncontrib[256];
        for (int p = 0; p < 256; p++) {
            const int px = tx * 16 + (p & 15), py = ty * 16 + (p >> 4);
            T[p] = 1.f; ncontrib[p] = 0; done[p] = (px >= W || py >= H);
        }
        uint16_t* reach = (uint16_t*)calloc(n, sizeof(uint16_t)); /* 4x4 sub-blocks of the tile the entry reaches */
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t id = point_list[r0 + k];
            const float* xy = means2D + 2 * (size_t)id; const float* co = conic_opacity + 4 * (size_t)id;
            uint16_t rsub = 0, lsub = 0, bsub = 0; /* reached / live (before this entry) / blending sub-blocks */
            for (int p = 0; p < 256; p++) if (!done[p]) lsub |= (uint16_t)(1u << (((p >> 4) / 4) * 4 + (p & 15) / 4));
            uint8_t newdone[256]; memcpy(newdone, done, 256);
            for (int p = 0; p < 256; p++) {
                const int lx = p & 15, ly = p >> 4;
                const float pixx = (float)(tx * 16 + lx), pixy = (float)(ty * 16 + ly);
                const float dx = xy[0] - pixx, dy = xy[1] - pixy;
                const float q = fmaf(co[0] * dx, dx, (co[2] * dy) * dy);
                const float power = fmaf(-0.5f, q, -((co[1] * dx) * dy));
                if (power > 0.f) continue;
                const float alpha = fminf(0.99f, co[3] * expf_(power));
                if (alpha < 1.0f / 255.0f) continue;
                const int s = (ly / 4) * 4 + lx / 4;
                rsub |= (uint16_t)(1u << s);
                if (done[p]) continue;
                const float tt = T[p] * (1 - alpha);
                if (tt < 0.0001f) { newdone[p] = 1; continue; }
                T[p] = tt; ncontrib[p] = k + 1; bsub |= (uint16_t)(1u << s);
            }
            reach[k] = rsub;
            for (int b = 0; b < 4; b++) { /* 8x8 block b = (by, bx): sub-blocks (2by..2by+1, 2bx..2bx+1) */
                const int by = b >> 1, bx = b & 1;
                const uint16_t m = (uint16_t)((3u << (2 * bx)) << (8 * by) | (3u << (2 * bx)) << (8 * by + 4));
                if (!(lsub & m)) continue;            /* block all done: the wave has left */
                if (rsub & m) fv++;
                if (rsub & lsub & m) fs++;
                if (bsub & m) fb++;
                /* coarser sub-blocks: 8 wide x 4 high (two per block), 4 wide x 8 high */
                for (int h = 0; h < 2; h++) {
                    const uint16_t m84 = (uint16_t)((3u << (2 * bx)) << (8 * by + 4 * h));
                    if ((rsub & m84) && (lsub & m84)) { f84++; break; }
                }
                for (int h = 0; h < 2; h++) {
                    const uint16_t m48 = (uint16_t)(((1u << (2 * bx + h)) << (8 * by)) | ((1u << (2 * bx + h)) << (8 * by + 4)));
                    if ((rsub & m48) && (lsub & m48)) { f48++; break; }
                }
            }
            memcpy(done, newdone, 256);
        }
        uint32_t hi = 0;
        for (int p = 0; p < 256; p++) if (ncontrib[p] > hi) hi = ncontrib[p];
        for (uint32_t k = 0; k < hi; k++) {
            uint16_t asub = 0; /* sub-blocks with an active lane for entry k */
            for (int p = 0; p < 256; p++) if (k < ncontrib[p]) asub |= (uint16_t)(1u << (((p >> 4) / 4) * 4 + (p & 15) / 4));
            for (int b = 0; b < 4; b++) {
                const int by = b >> 1, bx = b & 1;
                const uint16_t m = (uint16_t)((3u << (2 * bx)) << (8 * by) | (3u << (2 * bx)) << (8 * by + 4));
                if ((reach[k] & m) && (asub & m)) bv++;
                if (reach[k] & asub & m) bs++;
            }
        }
        free(reach);
    }
    out[0] = fv; out[1] = fs; out[2] = fb; out[3] = bv; out[4] = bs; out[5] = bb; out[6] = f84; out[7] = f48;
}
