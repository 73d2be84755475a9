/* TEST INFRASTRUCTURE ONLY -- body of the CPU restatement, instantiated twice by
 * me_oracle.c: once with PIXEL = uint8_t (8-bit) and once with uint16_t (>8-bit).
 * Required macros: PIXEL, FN(name), SUM2_T (packed accumulator type is not used:
 * the Hadamard transforms below are computed in plain int32, which is what the
 * reference's SWAR arithmetic evaluates to -- source/common/pixel.cpp:181-252).
 */

/* ---------------- pixel primitives: source/common/pixel.cpp ---------------- */

/* pixel.cpp:92-109 */
static int FN(sad)(const PIXEL *a, intptr_t sa, const PIXEL *b, intptr_t sb, int w, int h)
{
    int sum = 0;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++)
            sum += abs((int)a[x] - (int)b[x]);
    return sum;
}

/* 4x4 Hadamard of the difference block, sum of |coeff| (before the >>1);
 * pixel.cpp:201-227 */
static int FN(had4x4)(const PIXEL *a, intptr_t sa, const PIXEL *b, intptr_t sb)
{
    int d[4][4], t[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            d[i][j] = (int)a[i * sa + j] - (int)b[i * sb + j];
    for (int i = 0; i < 4; i++)
    {
        int s01 = d[i][0] + d[i][1], d01 = d[i][0] - d[i][1];
        int s23 = d[i][2] + d[i][3], d23 = d[i][2] - d[i][3];
        t[i][0] = s01 + s23; t[i][1] = s01 - s23;
        t[i][2] = d01 + d23; t[i][3] = d01 - d23;
    }
    int sum = 0;
    for (int j = 0; j < 4; j++)
    {
        int s01 = t[0][j] + t[1][j], d01 = t[0][j] - t[1][j];
        int s23 = t[2][j] + t[3][j], d23 = t[2][j] - t[3][j];
        sum += abs(s01 + s23) + abs(s01 - s23) + abs(d01 + d23) + abs(d01 - d23);
    }
    return sum;
}

/* satd table of pixel.cpp:817-841: widths that are multiples of 8 use 8x4 tiles
 * (two 4x4 transforms summed, then >>1, pixel.cpp:230-252), widths 4 and 12 use
 * 4x4 tiles (>>1 each, pixel.cpp:256-270). */
static int FN(satd)(const PIXEL *a, intptr_t sa, const PIXEL *b, intptr_t sb, int w, int h)
{
    int total = 0;
    if ((w & 7) == 0)
    {
        for (int r = 0; r < h; r += 4)
            for (int c = 0; c < w; c += 8)
                total += (FN(had4x4)(a + r * sa + c, sa, b + r * sb + c, sb) +
                          FN(had4x4)(a + r * sa + c + 4, sa, b + r * sb + c + 4, sb)) >> 1;
    }
    else
    {
        for (int r = 0; r < h; r += 4)
            for (int c = 0; c < w; c += 4)
                total += FN(had4x4)(a + r * sa + c, sa, b + r * sb + c, sb) >> 1;
    }
    return total;
}

/* 8x8 Hadamard abs-sum without rounding; pixel.cpp:290-325 */
static int FN(had8x8)(const PIXEL *a, intptr_t sa, const PIXEL *b, intptr_t sb)
{
    int m[8][8];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++)
            m[i][j] = (int)a[i * sa + j] - (int)b[i * sb + j];
    for (int pass = 0; pass < 2; pass++)
    {
        for (int i = 0; i < 8; i++)
        {
            int v[8];
            for (int j = 0; j < 8; j++) v[j] = pass ? m[j][i] : m[i][j];
            for (int s = 1; s < 8; s <<= 1)
                for (int j = 0; j < 8; j++)
                    if (!(j & s)) { int p = v[j], q = v[j + s]; v[j] = p + q; v[j + s] = p - q; }
            for (int j = 0; j < 8; j++) { if (pass) m[j][i] = v[j]; else m[i][j] = v[j]; }
        }
    }
    int sum = 0;
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++)
            sum += abs(m[i][j]);
    return sum;
}

/* sa8d_inter table, pixel.cpp:929-953 with the satd aliases of primitives.cpp:138-144 */
static int FN(sa8d_inter)(const PIXEL *a, intptr_t sa, const PIXEL *b, intptr_t sb, int w, int h)
{
    if ((w & 7) || (h & 7))
        return FN(satd)(a, sa, b, sb, w, h);
    int use16 = !(w & 15) && !(h & 15);
    int cost = 0;
    if (use16)
    {
        /* sa8d_16x16 rounds once per 16x16 (pixel.cpp:332-342) */
        for (int y = 0; y < h; y += 16)
            for (int x = 0; x < w; x += 16)
            {
                int s = 0;
                for (int k = 0; k < 4; k++)
                    s += FN(had8x8)(a + (y + (k >> 1) * 8) * sa + x + (k & 1) * 8, sa,
                                    b + (y + (k >> 1) * 8) * sb + x + (k & 1) * 8, sb);
                cost += (s + 2) >> 2;
            }
    }
    else
    {
        for (int y = 0; y < h; y += 8)
            for (int x = 0; x < w; x += 8)
                cost += (FN(had8x8)(a + y * sa + x, sa, b + y * sb + x, sb) + 2) >> 2;
    }
    return cost;
}

/* pixel.cpp:546-560 */
static void FN(pixelavg)(PIXEL *dst, intptr_t ds, const PIXEL *a, intptr_t sa, const PIXEL *b, intptr_t sb, int w, int h)
{
    for (int y = 0; y < h; y++, dst += ds, a += sa, b += sb)
        for (int x = 0; x < w; x++)
            dst[x] = (PIXEL)(((int)a[x] + (int)b[x] + 1) >> 1);
}

/* pixel.cpp:595-618: 2x downscale into the full-pel and the H / V / centre
 * half-pel planes, nested rounding averages */
static void FN(lowres_core)(const PIXEL *src0, PIXEL *d0, PIXEL *dh, PIXEL *dv, PIXEL *dc,
                            intptr_t ss, intptr_t ds, int w, int h)
{
#define AVG4(a, b, c, d) (PIXEL)((((((int)(a) + (int)(b) + 1) >> 1) + (((int)(c) + (int)(d) + 1) >> 1) + 1) >> 1))
    for (int y = 0; y < h; y++)
    {
        const PIXEL *s0 = src0 + 2 * y * ss, *s1 = s0 + ss, *s2 = s1 + ss;
        for (int x = 0; x < w; x++)
        {
            d0[y * ds + x] = AVG4(s0[2 * x], s1[2 * x], s0[2 * x + 1], s1[2 * x + 1]);
            dh[y * ds + x] = AVG4(s0[2 * x + 1], s1[2 * x + 1], s0[2 * x + 2], s1[2 * x + 2]);
            dv[y * ds + x] = AVG4(s1[2 * x], s2[2 * x], s1[2 * x + 1], s2[2 * x + 1]);
            dc[y * ds + x] = AVG4(s1[2 * x + 1], s2[2 * x + 1], s1[2 * x + 2], s2[2 * x + 2]);
        }
    }
#undef AVG4
}

/* ---------------- interpolation: source/common/ipfilter.cpp ---------------- */

static inline PIXEL FN(clip_pix)(int16_t v, int depth)
{
    int maxv = (1 << depth) - 1;
    if (v < 0) v = 0;
    if (v > maxv) v = (int16_t)maxv;
    return (PIXEL)v;
}

/* ipfilter.cpp:194-233; note the int16 cast BEFORE the clip (ipfilter.cpp:223) */
static void FN(luma_hpp)(const PIXEL *src, intptr_t ss, PIXEL *dst, intptr_t ds, int ci, int w, int h, int depth)
{
    const int16_t *c = orc_luma_filter[ci];
    src -= 3;
    for (int r = 0; r < h; r++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += (int)src[x + k] * c[k];
            dst[x] = FN(clip_pix)((int16_t)((sum + 32) >> 6), depth);
        }
}

/* ipfilter.cpp:280-319 */
static void FN(luma_vpp)(const PIXEL *src, intptr_t ss, PIXEL *dst, intptr_t ds, int ci, int w, int h, int depth)
{
    const int16_t *c = orc_luma_filter[ci];
    src -= 3 * ss;
    for (int r = 0; r < h; r++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += (int)src[x + k * ss] * c[k];
            dst[x] = FN(clip_pix)((int16_t)((sum + 32) >> 6), depth);
        }
}

/* ipfilter.cpp:235-278 */
static void FN(luma_hps)(const PIXEL *src, intptr_t ss, int16_t *dst, intptr_t ds, int ci, int isRowExt,
                         int w, int h, int depth)
{
    const int16_t *c = orc_luma_filter[ci];
    int headRoom = 14 - depth;
    int shift = 6 - headRoom;
    int offset = -(8192 << shift);
    int rows = h;
    src -= 3;
    if (isRowExt) { src -= 3 * ss; rows += 7; }
    for (int r = 0; r < rows; r++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += (int)src[x + k] * c[k];
            dst[x] = (int16_t)((sum + offset) >> shift);
        }
}

/* ipfilter.cpp:359-400 */
static void FN(luma_vsp)(const int16_t *src, intptr_t ss, PIXEL *dst, intptr_t ds, int ci, int w, int h, int depth)
{
    const int16_t *c = orc_luma_filter[ci];
    int headRoom = 14 - depth;
    int shift = 6 + headRoom;
    int offset = (1 << (shift - 1)) + (8192 << 6);
    src -= 3 * ss;
    for (int r = 0; r < h; r++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += (int)src[x + k * ss] * c[k];
            dst[x] = FN(clip_pix)((int16_t)((sum + offset) >> shift), depth);
        }
}

/* ---------------- motion search: source/encoder/motion.cpp ---------------- */

typedef struct
{
    const orc_me_ctx *c;
    PIXEL fenc[64 * 64];      /* FENC_STRIDE 64 cache, motion.cpp:116-142 */
    int w, h, part;
    intptr_t off;
    const PIXEL *fref;        /* ref_planes[0] + off */
    intptr_t stride;
    const uint16_t *cost;     /* centred */
    int mvpx, mvpy;
    orc_me_stats *st;
} FN(me_t);

static inline int FN(mvcost)(const FN(me_t) *m, int x, int y)
{
    /* bitcost.h:46: the sum of the two table entries is truncated to uint16 */
    return (uint16_t)(m->cost[x - m->mvpx] + m->cost[y - m->mvpy]);
}

static inline int FN(fpsad)(FN(me_t) *m, int mx, int my)
{
    if (m->st) { m->st->sad_pixels += (uint64_t)m->w * m->h; m->st->sad_calls++; }
    orc_trace((int64_t)(m->off + mx + my * m->stride));
    return FN(sad)(m->fenc, 64, m->fref + mx + my * m->stride, m->stride, m->w, m->h);
}

typedef int (*FN(cmp_t))(const PIXEL *, intptr_t, const PIXEL *, intptr_t, int, int);

/* motion.cpp:1651-1689 */
static int FN(subpel_cmp)(FN(me_t) *m, int qx, int qy, FN(cmp_t) cmp)
{
    int xFrac = qx & 3, yFrac = qy & 3;
    int mvX = qx >> 2, mvY = qy >> 2;
    int depth = m->c->depth;
    const PIXEL *fref = (const PIXEL*)m->c->ref_planes[0] + m->off + mvX + mvY * m->stride;
    if (m->st) { m->st->subpel_pixels += (uint64_t)m->w * m->h; m->st->subpel_calls++; }
    if (cmp == FN(sad))
        orc_trace((xFrac | yFrac) ? -1 : (int64_t)(m->off + mvX + mvY * m->stride));
    if ((xFrac | yFrac) == 0)
        return cmp(m->fenc, 64, fref, m->stride, m->w, m->h);
    PIXEL buf[64 * 64];
    if (yFrac == 0)
        FN(luma_hpp)(fref, m->stride, buf, 64, xFrac, m->w, m->h, depth);
    else if (xFrac == 0)
        FN(luma_vpp)(fref, m->stride, buf, 64, yFrac, m->w, m->h, depth);
    else
    {
        int16_t immed[64 * (64 + 8)];
        FN(luma_hps)(fref, m->stride, immed, m->w, xFrac, 1, m->w, m->h, depth);
        FN(luma_vsp)(immed + 3 * m->w, m->w, buf, 64, yFrac, m->w, m->h, depth);
    }
    return cmp(m->fenc, 64, buf, 64, m->w, m->h);
}

/* lowres.h:78-97: half-pel planes are precomputed, quarter-pel = rounding average
 * of the two nearest half-pel planes, always on an 8x8 block */
static int FN(lowres_cost)(FN(me_t) *m, int qx, int qy, FN(cmp_t) cmp)
{
    const orc_me_ctx *c = m->c;
    if (m->st) { m->st->subpel_pixels += (uint64_t)m->w * m->h; m->st->subpel_calls++; }
    int hpelA = (qy & 2) | ((qx & 2) >> 1);
    const PIXEL *frefA = (const PIXEL*)c->ref_planes[hpelA] + m->off + (qx >> 2) + (qy >> 2) * m->stride;
    if ((qx | qy) & 1)
    {
        PIXEL buf[8 * 8];
        int16_t bx = (int16_t)(qx + (qx & 1) * 2), by = (int16_t)(qy + (qy & 1) * 2);
        int hpelB = (by & 2) | ((bx & 2) >> 1);
        const PIXEL *frefB = (const PIXEL*)c->ref_planes[hpelB] + m->off + (bx >> 2) + (by >> 2) * m->stride;
        FN(pixelavg)(buf, 8, frefA, m->stride, frefB, m->stride, 8, 8);
        return cmp(m->fenc, 64, buf, 8, m->w, m->h);
    }
    return cmp(m->fenc, 64, frefA, m->stride, m->w, m->h);
}

static inline int FN(qcost)(FN(me_t) *m, int qx, int qy, FN(cmp_t) cmp)
{
    return m->c->is_lowres ? FN(lowres_cost)(m, qx, qy, cmp) : FN(subpel_cmp)(m, qx, qy, cmp);
}

#define S16(v) ((int16_t)(v))
#define IN_RANGE(x, y) ((x) >= mnx && (x) <= mxx && (y) >= mny && (y) <= mxy)
/* COST_MV (motion.cpp:196-202): full-pel candidate with strict-less update */
#define TRY_MV_AT(sx, sy, mx, my) do { int _x = (mx), _y = (my); \
        int _c = FN(fpsad)(m, (sx), (sy)) + FN(mvcost)(m, S16(_x << 2), S16(_y << 2)); \
        if (_c < bcost) { bcost = _c; bmx = S16(_x); bmy = S16(_y); } } while (0)
#define TRY_MV(mx, my) TRY_MV_AT(mx, my, mx, my)

/* StarPatternSearch, motion.cpp:307-547. Returns through the pointers. */
static void FN(star_pattern)(FN(me_t) *m, int mnx, int mny, int mxx, int mxy,
                             int16_t *pbmx, int16_t *pbmy, int *pbcost, int *pPointNr, int *pDistance,
                             int earlyExitIters, int merange)
{
    int16_t bmx = *pbmx, bmy = *pbmy;
    int bcost = *pbcost, bPointNr = *pPointNr, bDistance = *pDistance;
    const int16_t ox = bmx, oy = bmy;
    int saved = bcost, rounds = 0;

#define PT(mx, my, point, dist) do { int _x = S16(mx), _y = S16(my); \
        int _c = FN(fpsad)(m, _x, _y) + FN(mvcost)(m, S16(_x << 2), S16(_y << 2)); \
        if (_c < bcost) { bcost = _c; bmx = S16(_x); bmy = S16(_y); bPointNr = (point); bDistance = (dist); } } while (0)
#define ROUND_END() do { if (bcost < saved) rounds = 0; else if (++rounds >= earlyExitIters) goto done; } while (0)

    {
        const int16_t dist = 1;
        const int16_t top = S16(oy - dist), bottom = S16(oy + dist), left = S16(ox - dist), right = S16(ox + dist);
        /* the all-inside branch visits the same 4 points in the same order (motion.cpp:338-363) */
        if (top >= mny) PT(ox, top, 2, dist);
        if (left >= mnx) PT(left, oy, 4, dist);
        if (right <= mxx) PT(right, oy, 5, dist);
        if (bottom <= mxy) PT(ox, bottom, 7, dist);
        ROUND_END();
    }

    for (int16_t dist = 2; dist <= 8; dist <<= 1)
    {
        const int16_t top = S16(oy - dist), bottom = S16(oy + dist), left = S16(ox - dist), right = S16(ox + dist);
        const int16_t top2 = S16(oy - (dist >> 1)), bottom2 = S16(oy + (dist >> 1));
        const int16_t left2 = S16(ox - (dist >> 1)), right2 = S16(ox + (dist >> 1));
        saved = bcost;
        if (top >= mny && left >= mnx && right <= mxx && bottom <= mxy)
        {
            /* visit order of the two sad_x4 groups, motion.cpp:394-401 */
            PT(ox, top, 2, dist);
            PT(left2, top2, 1, dist >> 1);
            PT(right2, top2, 3, dist >> 1);
            PT(left, oy, 4, dist);
            PT(right, oy, 5, dist);
            PT(left2, bottom2, 6, dist >> 1);
            PT(right2, bottom2, 8, dist >> 1);
            PT(ox, bottom, 7, dist);
        }
        else
        {
            /* motion.cpp:405-442 */
            if (top >= mny) PT(ox, top, 2, dist);
            if (top2 >= mny)
            {
                if (left2 >= mnx) PT(left2, top2, 1, dist >> 1);
                if (right2 <= mxx) PT(right2, top2, 3, dist >> 1);
            }
            if (left >= mnx) PT(left, oy, 4, dist);
            if (right <= mxx) PT(right, oy, 5, dist);
            if (bottom2 <= mxy)
            {
                if (left2 >= mnx) PT(left2, bottom2, 6, dist >> 1);
                if (right2 <= mxx) PT(right2, bottom2, 8, dist >> 1);
            }
            if (bottom <= mxy) PT(ox, bottom, 7, dist);
        }
        ROUND_END();
    }

    for (int16_t dist = 16; dist <= (int16_t)merange; dist <<= 1)
    {
        const int16_t top = S16(oy - dist), bottom = S16(oy + dist), left = S16(ox - dist), right = S16(ox + dist);
        saved = bcost;
        if (top >= mny && left >= mnx && right <= mxx && bottom <= mxy)
        {
            PT(ox, top, 0, dist);
            PT(left, oy, 0, dist);
            PT(right, oy, 0, dist);
            PT(ox, bottom, 0, dist);
            for (int16_t index = 1; index < 4; index++)
            {
                int16_t posYT = S16(top + ((dist >> 2) * index)), posYB = S16(bottom - ((dist >> 2) * index));
                int16_t posXL = S16(ox - ((dist >> 2) * index)), posXR = S16(ox + ((dist >> 2) * index));
                PT(posXL, posYT, 0, dist);
                PT(posXR, posYT, 0, dist);
                PT(posXL, posYB, 0, dist);
                PT(posXR, posYB, 0, dist);
            }
        }
        else
        {
            if (top >= mny) PT(ox, top, 0, dist);
            if (left >= mnx) PT(left, oy, 0, dist);
            if (right <= mxx) PT(right, oy, 0, dist);
            if (bottom <= mxy) PT(ox, bottom, 0, dist);
            for (int16_t index = 1; index < 4; index++)
            {
                int16_t posYT = S16(top + ((dist >> 2) * index)), posYB = S16(bottom - ((dist >> 2) * index));
                int16_t posXL = S16(ox - ((dist >> 2) * index)), posXR = S16(ox + ((dist >> 2) * index));
                if (posYT >= mny)
                {
                    if (posXL >= mnx) PT(posXL, posYT, 0, dist);
                    if (posXR <= mxx) PT(posXR, posYT, 0, dist);
                }
                if (posYB <= mxy)
                {
                    if (posXL >= mnx) PT(posXL, posYB, 0, dist);
                    if (posXR <= mxx) PT(posXR, posYB, 0, dist);
                }
            }
        }
        ROUND_END();
    }
done:
#undef PT
#undef ROUND_END
    *pbmx = bmx; *pbmy = bmy; *pbcost = bcost; *pPointNr = bPointNr; *pDistance = bDistance;
}

/* motion.cpp:549-1146 */
static int FN(me_search)(const orc_me_ctx *c, intptr_t off, int w, int h, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp,
                         int numc, const orc_mv *mvc, int merange, orc_mv *out, orc_me_stats *st)
{
    static const int8_t hex2[8][2] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
    static const uint8_t mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
    static const int8_t square1[9][2] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
    static const int8_t hex4[16][2] = { {0,-4}, {0,4}, {-2,-3}, {2,-3}, {-4,-2}, {4,-2}, {-4,-1}, {4,-1},
                                        {-4,0}, {4,0}, {-4,1}, {4,1}, {-4,2}, {4,2}, {-2,3}, {2,3} };
    static const int8_t twopt[16][2] = { {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
                                         {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1} };
    /* SubpelWorkload, motion.cpp:53-63: hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd */
    static const int8_t workload[8][5] = { {1,4,0,4,0}, {1,4,1,4,0}, {1,4,1,4,1}, {2,4,1,4,1},
                                           {2,4,2,4,1}, {1,8,1,8,1}, {2,8,1,8,1}, {2,8,2,8,1} };

    FN(me_t) mm, *m = &mm;
    m->c = c; m->w = w; m->h = h; m->off = off; m->st = st;
    m->stride = c->ref_stride;
    m->fref = (const PIXEL*)c->ref_planes[0] + off;
    m->cost = c->cost + 32767;
    m->mvpx = qmvp.x; m->mvpy = qmvp.y; /* setMVP, bitcost.h:43 */
    {
        const PIXEL *src = (const PIXEL*)c->fenc_plane + off;
        for (int y = 0; y < h; y++)
            memcpy(m->fenc + 64 * y, src + y * c->fenc_stride, w * sizeof(PIXEL));
    }
    const int mnx = mvmin.x, mny = mvmin.y, mxx = mvmax.x, mxy = mvmax.y;
    const int16_t qmnx = S16(mnx << 2), qmny = S16(mny << 2), qmxx = S16(mxx << 2), qmxy = S16(mxy << 2);
    int costs[16];

#define CLIPQ_X(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define CLIPQ_Y(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
    /* clipped qpel predictor, motion.cpp:575-582 (no mvcost on the predictor itself) */
    int16_t pmx = CLIPQ_X(qmvp.x), pmy = CLIPQ_Y(qmvp.y);
    int16_t bpx = pmx, bpy = pmy;
    int bprecost = FN(qcost)(m, pmx, pmy, FN(sad));

    int16_t bmx = S16((pmx + 2) >> 2), bmy = S16((pmy + 2) >> 2);
    int bcost = bprecost;
    if ((pmx & 3) | (pmy & 3))
        bcost = FN(fpsad)(m, bmx, bmy) + FN(mvcost)(m, S16(bmx << 2), S16(bmy << 2));

    if (pmx | pmy)
    {
        int cost = FN(fpsad)(m, 0, 0) + FN(mvcost)(m, 0, 0);
        if (cost < bcost) { bcost = cost; bmx = 0; bmy = 0; }
    }

    for (int i = 0; i < numc; i++)
    {
        int16_t cx = CLIPQ_X(mvc[i].x), cy = CLIPQ_Y(mvc[i].y);
        if ((cx | cy) && !(cx == pmx && cy == pmy) && !(cx == bpx && cy == bpy))
        {
            int cost = FN(qcost)(m, cx, cy, FN(sad)) + FN(mvcost)(m, cx, cy);
            if (cost < bprecost) { bprecost = cost; bpx = cx; bpy = cy; }
        }
    }

    pmx = S16((pmx + 2) >> 2); pmy = S16((pmy + 2) >> 2);
    int16_t omx = bmx, omy = bmy;

    /* sad at (base + d) for a group, costs include mvcost of (cm + d) << 2 */
#define GROUP_COST(k, basex, basey, cmx, cmy, dx, dy) \
    costs[k] = FN(fpsad)(m, (basex) + (dx), (basey) + (dy)) + \
               FN(mvcost)(m, S16(S16((cmx) + (dx)) << 2), S16(S16((cmy) + (dy)) << 2))
#define LT1(a, b) do { if ((b) < (a)) (a) = (b); } while (0)

    switch (c->method)
    {
    case ORC_DIA:
    {
        /* motion.cpp:628-649 */
        bcost <<= 4;
        int i = merange;
        do
        {
            GROUP_COST(0, bmx, bmy, bmx, bmy, 0, -1);
            GROUP_COST(1, bmx, bmy, bmx, bmy, 0, 1);
            GROUP_COST(2, bmx, bmy, bmx, bmy, -1, 0);
            GROUP_COST(3, bmx, bmy, bmx, bmy, 1, 0);
            LT1(bcost, (costs[0] << 4) + 1);
            LT1(bcost, (costs[1] << 4) + 3);
            LT1(bcost, (costs[2] << 4) + 4);
            LT1(bcost, (costs[3] << 4) + 12);
            if (!(bcost & 15))
                break;
            bmx = S16(bmx - (((int32_t)((uint32_t)bcost << 28)) >> 30));
            bmy = S16(bmy - (((int32_t)((uint32_t)bcost << 30)) >> 30));
            bcost &= ~15;
        }
        while (--i && IN_RANGE(bmx, bmy));
        bcost >>= 4;
        break;
    }

    case ORC_HEX:
    {
me_hex2:
        /* motion.cpp:673-706: the six hexagon points are evaluated without a range check */
        GROUP_COST(0, bmx, bmy, bmx, bmy, -2, 0);
        GROUP_COST(1, bmx, bmy, bmx, bmy, -1, 2);
        GROUP_COST(2, bmx, bmy, bmx, bmy, 1, 2);
        bcost <<= 3;
        LT1(bcost, (costs[0] << 3) + 2);
        LT1(bcost, (costs[1] << 3) + 3);
        LT1(bcost, (costs[2] << 3) + 4);
        GROUP_COST(0, bmx, bmy, bmx, bmy, 2, 0);
        GROUP_COST(1, bmx, bmy, bmx, bmy, 1, -2);
        GROUP_COST(2, bmx, bmy, bmx, bmy, -1, -2);
        LT1(bcost, (costs[0] << 3) + 5);
        LT1(bcost, (costs[1] << 3) + 6);
        LT1(bcost, (costs[2] << 3) + 7);

        if (bcost & 7)
        {
            int dir = (bcost & 7) - 2;
            bmx = S16(bmx + hex2[dir + 1][0]); bmy = S16(bmy + hex2[dir + 1][1]);
            for (int i = (merange >> 1) - 1; i > 0 && IN_RANGE(bmx, bmy); i--)
            {
                GROUP_COST(0, bmx, bmy, bmx, bmy, hex2[dir + 0][0], hex2[dir + 0][1]);
                GROUP_COST(1, bmx, bmy, bmx, bmy, hex2[dir + 1][0], hex2[dir + 1][1]);
                GROUP_COST(2, bmx, bmy, bmx, bmy, hex2[dir + 2][0], hex2[dir + 2][1]);
                bcost &= ~7;
                LT1(bcost, (costs[0] << 3) + 1);
                LT1(bcost, (costs[1] << 3) + 2);
                LT1(bcost, (costs[2] << 3) + 3);
                if (!(bcost & 7))
                    break;
                dir += (bcost & 7) - 2;
                dir = mod6m1[dir + 1];
                bmx = S16(bmx + hex2[dir + 1][0]); bmy = S16(bmy + hex2[dir + 1][1]);
            }
        }
        bcost >>= 3;

        /* square refine, motion.cpp:709-722, also unchecked */
        int dir = 0;
        for (int k = 1; k <= 8; k++)
        {
            /* order of the two sad_x4 groups: (0,-1)(0,1)(-1,0)(1,0) then (-1,-1)(-1,1)(1,-1)(1,1) == square1[1..8] */
            GROUP_COST(0, bmx, bmy, bmx, bmy, square1[k][0], square1[k][1]);
            if (costs[0] < bcost) { bcost = costs[0]; dir = k; }
        }
        bmx = S16(bmx + square1[dir][0]); bmy = S16(bmy + square1[dir][1]);
        break;
    }

    case ORC_UMH:
    {
        /* motion.cpp:725-909. The SAD base of the X4 groups stays at the bmv the
         * case was entered with (pix_base is computed once, motion.cpp:732) while
         * the MV bookkeeping follows omv. */
        static const uint8_t range_mul[4][4] = { {3,3,4,4}, {3,4,4,4}, {4,4,4,5}, {4,4,5,6} };
        const int size_scale = (h * h) >> 4; /* motion.cpp:71-72 */
#define SAD_THRESH(v) (bcost < (((v) >> 4) * size_scale))
        const int16_t pbx = bmx, pby = bmy; /* pix_base */
        int ucost1, ucost2;
        int16_t cross_start = 1;
#define X4(d0x, d0y, d1x, d1y, d2x, d2y, d3x, d3y) do { \
            const int _dx[4] = { d0x, d1x, d2x, d3x }, _dy[4] = { d0y, d1y, d2y, d3y }; \
            for (int _k = 0; _k < 4; _k++) GROUP_COST(_k, pbx, pby, omx, omy, _dx[_k], _dy[_k]); \
            for (int _k = 0; _k < 4; _k++) \
                if (costs[_k] < bcost) { bcost = costs[_k]; bmx = S16(omx + _dx[_k]); bmy = S16(omy + _dy[_k]); } \
        } while (0)
#define DIA1(mx, my) do { omx = (mx); omy = (my); X4(0, -1, 0, 1, -1, 0, 1, 0); } while (0)
#define CROSS(start, x_max, y_max) do { \
            int16_t _i = (start); \
            if ((x_max) <= (mxx - omx < omx - mnx ? mxx - omx : omx - mnx)) \
                for (; _i < (x_max) - 2; _i += 4) X4(_i, 0, -_i, 0, _i + 2, 0, -_i - 2, 0); \
            for (; _i < (x_max); _i += 2) { \
                if (omx + _i <= mxx) TRY_MV(omx + _i, omy); \
                if (omx - _i >= mnx) TRY_MV(omx - _i, omy); } \
            _i = (start); \
            if ((y_max) <= (mxy - omy < omy - mny ? mxy - omy : omy - mny)) \
                for (; _i < (y_max) - 2; _i += 4) X4(0, _i, 0, -_i, 0, _i + 2, 0, -_i - 2); \
            for (; _i < (y_max); _i += 2) { \
                /* COST_MV(omv.x, omv.y + i) expands to fref + omv.x + omv.y + i * stride \
                 * (motion.cpp:199,301,303): the SAD is taken at (omv.x + omv.y, +-i) while \
                 * the MV bookkeeping uses (omv.x, omv.y +- i) */ \
                if (omy + _i <= mxy) TRY_MV_AT(omx + omy, _i, omx, omy + _i); \
                if (omy - _i >= mny) TRY_MV_AT(omx + omy, -_i, omx, omy - _i); } \
        } while (0)

        omx = bmx; omy = bmy;
        ucost1 = bcost;
        DIA1(pmx, pmy);
        if (pmx | pmy)
            DIA1(0, 0);

        ucost2 = bcost;
        if ((bmx | bmy) && !(bmx == pmx && bmy == pmy))
            DIA1(bmx, bmy);
        if (bcost == ucost2)
            cross_start = 3;

        omx = bmx; omy = bmy;
        if (bcost == ucost2 && SAD_THRESH(2000))
        {
            X4(0, -2, -1, -1, 1, -1, -2, 0);
            X4(2, 0, -1, 1, 1, 1, 0, 2);
            if (bcost == ucost1 && SAD_THRESH(500))
                break;
            if (bcost == ucost2)
            {
                int16_t range = (int16_t)((merange >> 1) | 1);
                CROSS(3, range, range);
                X4(-1, -2, 1, -2, -2, -1, 2, -1);
                X4(-2, 1, 2, 1, -1, 2, 1, 2);
                if (bcost == ucost2)
                    break;
                cross_start = S16(range + 2);
            }
        }

        if (numc)
        {
            int mvd, denom = 1;
            if (numc == 1)
            {
                if (w == 64 && h == 64) mvd = 25;
                else mvd = abs(qmvp.x - mvc[0].x) + abs(qmvp.y - mvc[0].y);
            }
            else
            {
                denom = numc - 1;
                mvd = 0;
                if (!(w == 64 && h == 64))
                {
                    mvd = abs(qmvp.x - mvc[0].x) + abs(qmvp.y - mvc[0].y);
                    denom++;
                }
                for (int i = 0; i < numc - 1; i++)
                    mvd += abs(mvc[i].x - mvc[i + 1].x) + abs(mvc[i].y - mvc[i + 1].y);
            }
            int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
            int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            merange = (merange * range_mul[mvd_ctx][sad_ctx]) >> 2;
        }

        CROSS(cross_start, merange, merange >> 1);
        X4(-2, -2, -2, 2, 2, -2, 2, 2);

        /* hexagon grid, motion.cpp:826-905 */
        omx = bmx; omy = bmy;
        uint16_t i = 1;
        do
        {
            int lim = mxx - omx;
            if (omx - mnx < lim) lim = omx - mnx;
            if (mxy - omy < lim) lim = mxy - omy;
            if (omy - mny < lim) lim = omy - mny;
            if (4 * i > lim)
            {
                for (int j = 0; j < 16; j++)
                {
                    int16_t mx = S16(omx + S16(hex4[j][0] * (int16_t)i)), my = S16(omy + S16(hex4[j][1] * (int16_t)i));
                    if (IN_RANGE(mx, my))
                        TRY_MV(mx, my);
                }
            }
            else
            {
                /* all 16 points measured first, then compared in table order */
                int dirx = 0, diry = 0, found = 0;
                for (int j = 0; j < 16; j++)
                {
                    int px = omx + hex4[j][0] * i, py = omy + hex4[j][1] * i;
                    /* p_cost_omvx[x*4*i] == m_cost_mvx[(omv.x + x*i) * 4] without the int16 wrap of MV << 2 */
                    costs[j] = FN(fpsad)(m, px, py) +
                               (int)m->cost[px * 4 - m->mvpx] + (int)m->cost[py * 4 - m->mvpy];
                }
                for (int j = 0; j < 16; j++)
                    if (costs[j] < bcost) { bcost = costs[j]; dirx = hex4[j][0]; diry = hex4[j][1]; found = 1; }
                if (found)
                {
                    bmx = S16(omx + i * dirx);
                    bmy = S16(omy + i * diry);
                }
            }
        }
        while (++i <= merange >> 2);
        if (IN_RANGE(bmx, bmy))
            goto me_hex2;
        break;
#undef SAD_THRESH
#undef X4
#undef DIA1
#undef CROSS
    }

    case ORC_STAR:
    {
        /* motion.cpp:911-1019 */
        int bPointNr = 0, bDistance = 0;
        FN(star_pattern)(m, mnx, mny, mxx, mxy, &bmx, &bmy, &bcost, &bPointNr, &bDistance, 3, merange);
        if (bDistance == 1)
        {
            if (bPointNr)
            {
                int saved = bcost;
                int16_t x1 = S16(bmx + twopt[(bPointNr - 1) * 2][0]), y1 = S16(bmy + twopt[(bPointNr - 1) * 2][1]);
                int16_t x2 = S16(bmx + twopt[(bPointNr - 1) * 2 + 1][0]), y2 = S16(bmy + twopt[(bPointNr - 1) * 2 + 1][1]);
                if (IN_RANGE(x1, y1)) TRY_MV(x1, y1);
                if (IN_RANGE(x2, y2)) TRY_MV(x2, y2);
                if (bcost == saved)
                    break;
            }
            else
                break;
        }

        if (bDistance > 5)
        {
            /* raster refinement, step 5; the 4th point of each full group of four is
             * costed with tmv << 3 instead of << 2 (motion.cpp:975) */
            for (int16_t ty = S16(mny); ty <= mxy; ty = S16(ty + 5))
            {
                for (int16_t tx = S16(mnx); tx <= mxx; tx = S16(tx + 5))
                {
                    if (tx + 15 <= mxx)
                    {
                        for (int k = 0; k < 4; k++)
                            costs[k] = FN(fpsad)(m, tx + 5 * k, ty);
                        for (int k = 0; k < 4; k++)
                        {
                            int sh = (k == 3) ? 3 : 2;
                            int cst = costs[k] + FN(mvcost)(m, S16(tx << sh), S16(ty << sh));
                            if (cst < bcost) { bcost = cst; bmx = tx; bmy = ty; }
                            if (k < 3) tx = S16(tx + 5);
                        }
                    }
                    else
                        TRY_MV(tx, ty);
                }
            }
        }

        while (bDistance > 0)
        {
            bDistance = 0;
            bPointNr = 0;
            FN(star_pattern)(m, mnx, mny, mxx, mxy, &bmx, &bmy, &bcost, &bPointNr, &bDistance, 32, merange);
            if (bDistance == 1)
            {
                if (!bPointNr)
                    break;
                int16_t x1 = S16(bmx + twopt[(bPointNr - 1) * 2][0]), y1 = S16(bmy + twopt[(bPointNr - 1) * 2][1]);
                int16_t x2 = S16(bmx + twopt[(bPointNr - 1) * 2 + 1][0]), y2 = S16(bmy + twopt[(bPointNr - 1) * 2 + 1][1]);
                if (IN_RANGE(x1, y1)) TRY_MV(x1, y1);
                if (IN_RANGE(x2, y2)) TRY_MV(x2, y2);
                break;
            }
        }
        break;
    }

    case ORC_FULL:
    {
        /* motion.cpp:1021-1056: raster order y then x, strict-less update */
        for (int ty = mny; ty <= mxy; ty++)
            for (int tx = mnx; tx <= mxx; tx++)
                TRY_MV(tx, ty);
        break;
    }
    default:
        return -1;
    }

    (void)omx; (void)omy;
    int16_t qx, qy;
    if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
    else { qx = S16(bmx << 2); qy = S16(bmy << 2); }

    const int8_t *wl = workload[c->subme];
    if (!bcost)
    {
        /* motion.cpp:1073-1076 */
    }
    else if (c->is_lowres)
    {
        /* motion.cpp:1077-1099 */
        int bdir = 0;
        for (int i = 1; i <= wl[1]; i++)
        {
            int16_t tx = S16(qx + square1[i][0] * 2), ty = S16(qy + square1[i][1] * 2);
            int cost = FN(lowres_cost)(m, tx, ty, FN(sad)) + FN(mvcost)(m, tx, ty);
            if (cost < bcost) { bcost = cost; bdir = i; }
        }
        qx = S16(qx + square1[bdir][0] * 2); qy = S16(qy + square1[bdir][1] * 2);
        bcost = FN(lowres_cost)(m, qx, qy, FN(satd)) + FN(mvcost)(m, qx, qy);
        bdir = 0;
        for (int i = 1; i <= wl[3]; i++)
        {
            int16_t tx = S16(qx + square1[i][0]), ty = S16(qy + square1[i][1]);
            int cost = FN(lowres_cost)(m, tx, ty, FN(satd)) + FN(mvcost)(m, tx, ty);
            if (cost < bcost) { bcost = cost; bdir = i; }
        }
        qx = S16(qx + square1[bdir][0]); qy = S16(qy + square1[bdir][1]);
    }
    else
    {
        /* motion.cpp:1100-1141 */
        FN(cmp_t) hpelcomp = FN(sad);
        if (wl[4])
        {
            bcost = FN(subpel_cmp)(m, qx, qy, FN(satd)) + FN(mvcost)(m, qx, qy);
            hpelcomp = FN(satd);
        }
        for (int iter = 0; iter < wl[0]; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl[1]; i++)
            {
                int16_t tx = S16(qx + square1[i][0] * 2), ty = S16(qy + square1[i][1] * 2);
                int cost = FN(subpel_cmp)(m, tx, ty, hpelcomp) + FN(mvcost)(m, tx, ty);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            qx = S16(qx + square1[bdir][0] * 2); qy = S16(qy + square1[bdir][1] * 2);
        }
        if (!wl[4])
            bcost = FN(subpel_cmp)(m, qx, qy, FN(satd)) + FN(mvcost)(m, qx, qy);
        for (int iter = 0; iter < wl[2]; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl[3]; i++)
            {
                int16_t tx = S16(qx + square1[i][0]), ty = S16(qy + square1[i][1]);
                int cost = FN(subpel_cmp)(m, tx, ty, FN(satd)) + FN(mvcost)(m, tx, ty);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            qx = S16(qx + square1[bdir][0]); qy = S16(qy + square1[bdir][1]);
        }
    }
    out->x = qx; out->y = qy;
    return bcost;
#undef GROUP_COST
#undef LT1
#undef CLIPQ_X
#undef CLIPQ_Y
}


/* ---------------- Rockchip CIME / RIME / FME: motion.cpp:1220-1648 ---------------- */

/* 4-tap kernels used by the RK fractional search: g_chromaFilter[frac * 2]
 * (motion.cpp:1455-1490, table TComRom.cpp:183-193) */
static const int16_t FN(rk_taps)[4][4] = { { 0, 64, 0, 0 }, { -4, 54, 16, -2 }, { -4, 36, 36, -4 }, { -2, 16, 54, -4 } };

/* RK subpelCompare(pfref, ...) evaluated directly on the reference plane; `r` is the
 * integer position (qmv >> 2) of the block. motion.cpp:1453-1494 with
 * InterpHoriz :1495, InterpVert :1534, FilterHorizontal :1573, FilterVertical :1608 */
static int FN(rk_subpel_satd)(const PIXEL *fenc64, const PIXEL *r, intptr_t stride, int qx, int qy, int w, int h, int depth)
{
    const int xf = qx & 3, yf = qy & 3;
    PIXEL fe[64 * 64];
    for (int y = 0; y < h; y++) memcpy(fe + y * w, fenc64 + y * 64, w * sizeof(PIXEL)); /* pfenc, stride = width */
    if ((xf | yf) == 0)
        return FN(satd)(fe, w, r, stride, w, h);
    PIXEL buf[64 * 64];
    const int16_t *ch = FN(rk_taps)[xf], *cv = FN(rk_taps)[yf];
    if (yf == 0)
    {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)r[y * stride + x - 1 + k] * ch[k];
                buf[y * 64 + x] = FN(clip_pix)((int16_t)((sum + 32) >> 6), depth);
            }
    }
    else if (xf == 0)
    {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)r[(y - 1 + k) * stride + x] * cv[k];
                buf[y * 64 + x] = FN(clip_pix)((int16_t)((sum + 32) >> 6), depth);
            }
    }
    else
    {
        int16_t im[64 * (64 + 8)];
        const int headRoom = 14 - depth, shiftH = 6 - headRoom, offH = -(8192 << shiftH);
        const int shiftV = 6 + headRoom, offV = (1 << (shiftV - 1)) + (8192 << 6);
        for (int y = 0; y < h + 3; y++)
            for (int x = 0; x < w; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)r[(y - 1) * stride + x - 1 + k] * ch[k];
                im[y * w + x] = (int16_t)((sum + offH) >> shiftH);
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)im[(y + k) * w + x] * cv[k];
                buf[y * 64 + x] = FN(clip_pix)((int16_t)((sum + offV) >> shiftV), depth);
            }
    }
    return FN(satd)(fe, w, buf, 64, w, h);
}

static inline int FN(rk_mvcost3)(const uint16_t *cost, const orc_rk_ctu *j, int x, int y)
{
    /* min over the three cost predictors, each sum truncated to uint16 (motion.cpp:1284-1292) */
    int best = 0x7fffffff;
    for (int k = 0; k < 3; k++)
    {
        int c = (uint16_t)(cost[x - j->mvp[k][0]] + cost[y - j->mvp[k][1]]);
        if (c < best) best = c;
    }
    return j->have_mvd ? best : 0;
}

static void FN(rk_ctu)(const orc_me_ctx *c, const int16_t *ds, intptr_t ds_stride, int pic_w, int pic_h,
                       const orc_rk_ctu *j, orc_rk_result *out)
{
    const uint16_t *cost = c->cost + 32767;
    const PIXEL *fplane = (const PIXEL*)c->fenc_plane;
    const PIXEL *rplane = (const PIXEL*)c->ref_planes[0];
    const intptr_t fs = c->fenc_stride, rs = c->ref_stride;
    memset(out, 0, sizeof(*out));

    /* ---- CIME: 4x4-sum downsampled CTU against the downsampled ORIGINAL reference,
     * full search over [mvmin / 4, mvmax / 4] (motion.cpp:1272-1303) */
    int16_t ctuDS[16 * 16];
    {
        const PIXEL *src = fplane + (intptr_t)j->ctu_y * fs + j->ctu_x; /* the RK build caches 64x64 at the CU offset */
        for (int i = 0; i < 16; i++)
            for (int k = 0; k < 16; k++)
            {
                int16_t sum = 0;
                for (int m = 0; m < 4; m++)
                    for (int n = 0; n < 4; n++) sum = (int16_t)(sum + src[(i * 4 + m) * fs + k * 4 + n]);
                ctuDS[i * 16 + k] = sum;
            }
    }
    const int16_t *refDS = ds + j->ctu_x / 4 + (intptr_t)(j->ctu_y / 4) * ds_stride;
    const int cw = j->valid_w / 4, chh = j->valid_h / 4;
    int bcost = 0x7fffffff;
    int16_t bx = 0, by = 0;
    for (int16_t ty = (int16_t)(j->mvmin[1] / 4); ty <= j->mvmax[1] / 4; ty++)
        for (int16_t tx = (int16_t)(j->mvmin[0] / 4); tx <= j->mvmax[0] / 4; tx++)
        {
            int sad = 0;
            const int16_t *p = refDS + tx + (intptr_t)ty * ds_stride;
            /* the CTU block is indexed with the VALID width as its row pitch (motion.cpp:1227) */
            for (int i = 0; i < chh; i++)
                for (int k = 0; k < cw; k++) sad += abs(p[k + i * ds_stride] - ctuDS[k + i * cw]);
            int cst = sad + FN(rk_mvcost3)(cost, j, tx, ty);
            if (cst < bcost) { bcost = cst; bx = tx; by = ty; }
        }
    int16_t pmv[3][2];
    pmv[0][0] = (int16_t)(bx * 4); pmv[0][1] = (int16_t)(by * 4);
    pmv[1][0] = j->pmv[0][0]; pmv[1][1] = j->pmv[0][1];
    pmv[2][0] = j->pmv[1][0]; pmv[2][1] = j->pmv[1][1];
    out->cime.mv[0] = pmv[0][0]; out->cime.mv[1] = pmv[0][1]; out->cime.cost = bcost;

    /* ---- per CU: RIME (3 windows of 16 x 12 full-pel points) then FME (7 x 7 qpel points) */
    for (int idx = 0; idx < 85; idx++)
    {
        int size, ox, oy;
        if (idx < 64) { size = 8; ox = (idx & 7) * 8; oy = (idx >> 3) * 8; }
        else if (idx < 80) { size = 16; ox = ((idx - 64) & 3) * 16; oy = ((idx - 64) >> 2) * 16; }
        else if (idx < 84) { size = 32; ox = ((idx - 80) & 1) * 32; oy = ((idx - 80) >> 1) * 32; }
        else { size = 64; ox = 0; oy = 0; }
        if (j->ctu_x + ox + size > pic_w || j->ctu_y + oy + size > pic_h)
        {
            out->rime[idx].cost = -1; out->fme[idx].cost = -1;   /* CU outside the picture (TEncCu.cpp:475-493) */
            continue;
        }
        const intptr_t off = (intptr_t)(j->ctu_y + oy) * fs + j->ctu_x + ox;
        const intptr_t roff = (intptr_t)(j->ctu_y + oy) * rs + j->ctu_x + ox;
        PIXEL fenc[64 * 64];
        for (int y = 0; y < size; y++) memcpy(fenc + 64 * y, fplane + off + y * fs, size * sizeof(PIXEL));
        bcost = 0x7fffffff;
        int16_t mx = 0, my = 0;
        for (int p = 0; p < 3; p++)
            for (int16_t i = -6; i < 6; i++)
                for (int16_t k = -8; k < 8; k++)
                {
                    int16_t tx = (int16_t)(pmv[p][0] + k), ty = (int16_t)(pmv[p][1] + i);
                    int cst = FN(sad)(fenc, 64, rplane + roff + tx + (intptr_t)ty * rs, rs, size, size) +
                              FN(rk_mvcost3)(cost, j, tx, ty);
                    if (cst < bcost) { bcost = cst; mx = tx; my = ty; }
                }
        out->rime[idx].mv[0] = mx; out->rime[idx].mv[1] = my; out->rime[idx].cost = bcost;

        const int16_t BX = (int16_t)(mx << 2), BY = (int16_t)(my << 2);
        bcost = 0x7fffffff;
        int16_t fx = 0, fy = 0;
        for (int16_t i = -3; i <= 3; i++)       /* x offset outer, y inner (motion.cpp:1377-1381) */
            for (int16_t k = -3; k <= 3; k++)
            {
                int16_t qx = (int16_t)(BX + i), qy = (int16_t)(BY + k);
                const PIXEL *r = rplane + roff + (qx >> 2) + (intptr_t)(qy >> 2) * rs;
                int cst = FN(rk_subpel_satd)(fenc, r, rs, qx, qy, size, size, c->depth) + FN(rk_mvcost3)(cost, j, qx, qy);
                if (cst < bcost) { bcost = cst; fx = qx; fy = qy; }
            }
        out->fme[idx].mv[0] = fx; out->fme[idx].mv[1] = fy; out->fme[idx].cost = bcost;
    }
}

#undef S16
#undef IN_RANGE
#undef TRY_MV
#undef TRY_MV_AT


/* ---------------- lookahead frame cost: slicetype.cpp:1401-1582 besides the search ---------------- */

/* The reference prediction edges of one 8x8 lowres CU (slicetype.cpp:1495-1524):
 * e[0] = the 9 samples above / left of the block starting at the corner, the last one
 * repeated up to index 16; e[1] = the same smoothed with [1 2 1], ends unfiltered. */
static void FN(la_edges)(const PIXEL *cur, intptr_t stride, int above[2][17], int left[2][17])
{
    for (int i = 0; i <= 8; i++)
    {
        above[0][i] = cur[-1 - stride + i];
        left[0][i] = cur[-1 - stride + i * stride];
    }
    for (int i = 9; i <= 16; i++) { above[0][i] = above[0][8]; left[0][i] = left[0][8]; }
    above[1][0] = above[0][0]; above[1][16] = above[0][16];
    left[1][0] = left[0][0]; left[1][16] = left[0][16];
    for (int i = 1; i < 16; i++)
    {
        above[1][i] = (above[0][i - 1] + 2 * above[0][i] + above[0][i + 1] + 2) >> 2;
        left[1][i] = (left[0][i - 1] + 2 * left[0][i] + left[0][i + 1] + 2) >> 2;
    }
}

/* Sample (row k, column l) of entry `mode` of the reference's prediction buffer
 * (slicetype.cpp:1529-1533): 0 = DC with edge smoothing (intrapred.cpp:45-104), 1 = planar
 * from the smoothed edges (:106-154), 2..34 = angular (:156-259) in the orientation
 * all_angs_pred_c leaves them in (:262-289): modes below 18 stay transposed. */
static int FN(la_pred)(int mode, int k, int l, int above[2][17], int left[2][17], int depth)
{
    static const int angTable[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
    static const int invAngTable[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
    const int maxv = (1 << depth) - 1;
    if (mode == 0)
    {
        int sum = 8;
        for (int i = 1; i <= 8; i++) sum += above[0][i] + left[0][i];
        const int dc = sum / 16;
        if (k == 0 && l == 0) return (above[0][1] + left[0][1] + 2 * dc + 2) >> 2;
        if (k == 0) return (above[0][l + 1] + 3 * dc + 2) >> 2;
        if (l == 0) return (left[0][k + 1] + 3 * dc + 2) >> 2;
        return dc;
    }
    if (mode == 1)
    {
        const int *a = above[1] + 1, *lf = left[1] + 1;
        return ((lf[k] << 3) + 8 + (l + 1) * (int16_t)(a[8] - lf[k]) + (a[l] << 3) + (k + 1) * (int16_t)(lf[8] - a[l])) >> 4;
    }
    const int f = (mode == 2 || mode == 18 || mode == 34);   /* IntraFilterType, 8x8 row (intrapred.cpp:31-40) */
    const int ver = mode >= 18;
    const int *refMain = ver ? above[f] : left[f], *refSide = ver ? left[f] : above[f];
    int ang = ver ? mode - 26 : -(mode - 10);
    const int absIdx = abs(ang);
    const int invAngle = invAngTable[absIdx];
    ang = ang < 0 ? -angTable[absIdx] : angTable[absIdx];
    if (ang == 0)
    {
        int v = refMain[l + 1];
        if (l == 0) /* edge filter, bFilter = (cuSize <= 16) */
        {
            int16_t t = (int16_t)(v + ((refSide[k + 1] - refSide[0]) >> 1));
            v = t < 0 ? 0 : (t > maxv ? maxv : t);
        }
        return v;
    }
    const int deltaPos = (k + 1) * ang, deltaInt = deltaPos >> 5, fract = deltaPos & 31;
    /* negative indices of the main edge are projections of the side edge (:181-195) */
#define MAINREF(i) ((i) >= 0 ? refMain[i] : refSide[(128 + -(i) * invAngle) >> 8])
    const int i0 = l + deltaInt + 1;
    if (!fract) return MAINREF(i0);
    return ((32 - fract) * MAINREF(i0) + fract * MAINREF(i0 + 1) + 16) >> 5;
#undef MAINREF
}

/* slicetype.cpp:1491-1551: least SATD over the 35 predictions of an 8x8 block */
static int FN(la_intra_cost)(const PIXEL *cur, intptr_t stride, int depth)
{
    int above[2][17], left[2][17];
    PIXEL src[64], srct[64], pred[64];
    FN(la_edges)(cur, stride, above, left);
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) { src[y * 8 + x] = cur[y * stride + x]; srct[x * 8 + y] = cur[y * stride + x]; }
    int icost = 1 << 28;
    for (int mode = 0; mode < 35; mode++)
    {
        for (int k = 0; k < 8; k++)
            for (int l = 0; l < 8; l++) pred[k * 8 + l] = (PIXEL)FN(la_pred)(mode, k, l, above, left, depth);
        const int cost = FN(satd)((mode >= 2 && mode < 18) ? srct : src, 8, pred, 8, 8, 8);
        if (cost < icost) icost = cost;
    }
    return icost;
}

/* lowres.h:56-76 into an 8x8 buffer (always copied: the caller only averages it) */
static void FN(la_mc)(const void *const planes[4], intptr_t off, intptr_t stride, int qx, int qy, PIXEL *buf)
{
    const int hpelA = (qy & 2) | ((qx & 2) >> 1);
    const PIXEL *a = (const PIXEL*)planes[hpelA] + off + (qx >> 2) + (qy >> 2) * stride;
    if ((qx | qy) & 1)
    {
        const int16_t bx = (int16_t)(qx + (qx & 1) * 2), by = (int16_t)(qy + (qy & 1) * 2);
        const int hpelB = (by & 2) | ((bx & 2) >> 1);
        const PIXEL *b = (const PIXEL*)planes[hpelB] + off + (bx >> 2) + (by >> 2) * stride;
        FN(pixelavg)(buf, 8, a, stride, b, stride, 8, 8);
    }
    else
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) buf[y * 8 + x] = a[y * stride + x];
}

static void FN(la_frame_cost)(int depth, const void *fenc0, const void *const ref0[4], const void *const ref1[4],
                              intptr_t stride, int wcu, int hcu, int d0, int d1, int intra_calculated,
                              const int32_t *inv_qscale, const orc_mv *mvs0, const int32_t *mvcosts0,
                              const orc_mv *mvs1, const int32_t *mvcosts1, int32_t *intra_cost, uint16_t *lowres_costs,
                              int32_t *row_satds_intra, int32_t *row_satds, int64_t totals[5])
{
    const PIXEL *fenc = (const PIXEL*)fenc0;
    const int bidir = d1 > 0, inter = (d0 + d1) > 0;
    uint64_t costEst = 0, costEstAq = 0, costIntraAq = 0;
    int costIntra = 0, intraMbs = 0;
    for (int cuy = hcu - 1; cuy >= 0; cuy--)
    {
        if (!intra_calculated) row_satds_intra[cuy] = 0;
        if (inter) row_satds[cuy] = 0;
        for (int cux = wcu - 1; cux >= 0; cux--)
        {
            const int cuXY = cux + cuy * wcu;
            const intptr_t off = 8 * cux + 8 * cuy * stride;
            const int scored = (cux > 0 && cux < wcu - 1 && cuy > 0 && cuy < hcu - 1) || wcu <= 2 || hcu <= 2;
            int bcost = 1 << 28, listused = 0;   /* MotionEstimate::COST_MAX, motion.h:64 */
            if (inter)
            {
                if (mvcosts0[cuXY] < bcost) { bcost = mvcosts0[cuXY]; listused = 1; }
                if (bidir)
                {
                    if (mvcosts1[cuXY] < bcost) { bcost = mvcosts1[cuXY]; listused = 2; }
                    PIXEL b0[64], b1[64], avg[64];
                    FN(la_mc)(ref0, off, stride, mvs0[cuXY].x, mvs0[cuXY].y, b0);
                    FN(la_mc)(ref1, off, stride, mvs1[cuXY].x, mvs1[cuXY].y, b1);
                    FN(pixelavg)(avg, 8, b0, 8, b1, 8, 8, 8);
                    int bicost = FN(satd)(fenc + off, stride, avg, 8, 8, 8);
                    if (bicost < bcost) { bcost = bicost; listused = 3; }
                    FN(pixelavg)(avg, 8, (const PIXEL*)ref0[0] + off, stride, (const PIXEL*)ref1[0] + off, stride, 8, 8);
                    bicost = FN(satd)(fenc + off, stride, avg, 8, 8, 8);
                    if (bicost < bcost) { bcost = bicost; listused = 3; }
                }
            }
            if (!intra_calculated)
            {
                const int icost = FN(la_intra_cost)(fenc + off, stride, depth);
                intra_cost[cuXY] = icost;
                row_satds_intra[cuy] += icost;
                if (scored)
                {
                    costIntra += icost;
                    if (inv_qscale) costIntraAq += (icost * inv_qscale[cuXY] + 128) >> 8;
                }
            }
            if (!bidir && intra_cost[cuXY] < bcost)
            {
                if (scored) intraMbs++;
                bcost = intra_cost[cuXY];
                listused = 0;
            }
            if (inter)
            {
                row_satds[cuy] += bcost;
                if (scored)
                {
                    costEst += bcost;
                    if (inv_qscale) costEstAq += (bcost * inv_qscale[cuXY] + 128) >> 8;
                }
            }
            lowres_costs[cuXY] = (uint16_t)((bcost < 16383 ? bcost : 16383) | (listused << 14));
        }
    }
    totals[0] = (int64_t)costEst; totals[1] = (int64_t)costEstAq; totals[2] = costIntra;
    totals[3] = (int64_t)costIntraAq; totals[4] = intraMbs;
}


/* ---------------- explicit weighted prediction ---------------- */

/* weight_pp_c, pixel.cpp:512-529: the sample goes through the 14-bit intermediate as an int16 */
static inline PIXEL FN(weight_px)(PIXEL v, int w0, int round, int shift, int offset, int depth)
{
    const int16_t val = (int16_t)(v << (14 - depth));
    int r = ((w0 * val + round) >> shift) + offset;
    const int maxv = (1 << depth) - 1;
    return (PIXEL)(r < 0 ? 0 : (r > maxv ? maxv : r));
}

static void FN(weight_pp)(const PIXEL *src, PIXEL *dst, intptr_t ss, intptr_t ds, int width, int height,
                          int w0, int round, int shift, int offset, int depth)
{
    for (int y = 0; y < height; y++, src += ss, dst += ds)
        for (int x = 0; x < width; x++) dst[x] = FN(weight_px)(src[x], w0, round, shift, offset, depth);
}

/* MotionReference::init + applyWeight over a whole picture (reference.cpp:39-71, :211-256): the
 * picture area weighted, then replicated into the margins on all sides */
static void FN(apply_weight)(const PIXEL *src_origin, PIXEL *dst_origin, intptr_t stride, int width, int height,
                             int mx, int my, int inputWeight, int log2Denom, int inputOffset, int depth)
{
    const int offset = inputOffset * (1 << (depth - 8));
    const int shift = log2Denom + (14 - depth);
    const int round = shift ? 1 << (shift - 1) : 0;
    FN(weight_pp)(src_origin, dst_origin, stride, stride, width, height, inputWeight, round, shift, offset, depth);
    for (int y = 0; y < height; y++)
    {
        PIXEL *r = dst_origin + y * stride;
        for (int x = 1; x <= mx; x++) { r[-x] = r[0]; r[width - 1 + x] = r[width - 1]; }
    }
    for (int y = 1; y <= my; y++)
    {
        memcpy(dst_origin - mx - y * stride, dst_origin - mx, (width + 2 * mx) * sizeof(PIXEL));
        memcpy(dst_origin - mx + (height - 1 + y) * stride, dst_origin - mx + (height - 1) * stride, (width + 2 * mx) * sizeof(PIXEL));
    }
}

/* CostEstimate::weightCostLuma, slicetype.cpp:1223-1256 */
static uint32_t FN(weight_cost_luma)(const PIXEL *fenc, const PIXEL *src, intptr_t stride, int wcu, int hcu,
                                     const int32_t *intra_cost, int use_wp, int scale, int denom, int inputOffset, int depth)
{
    const int correction = 14 - depth;
    uint32_t cost = 0;
    int mb = 0;
    for (int y = 0; y < hcu * 8; y += 8)
        for (int x = 0; x < wcu * 8; x += 8, mb++)
        {
            PIXEL blk[64];
            for (int j = 0; j < 8; j++)
                for (int i = 0; i < 8; i++)
                {
                    const PIXEL v = src[(y + j) * stride + x + i];
                    blk[j * 8 + i] = use_wp ? FN(weight_px)(v, scale, 1 << (denom - 1 + correction), denom + correction,
                                                            inputOffset << (depth - 8), depth) : v;
                }
            const int satd = FN(satd)(blk, 8, fenc + y * stride + x, stride, 8, 8);
            cost += satd < intra_cost[mb] ? satd : intra_cost[mb];
        }
    return cost;
}


/* ---------------- bi-prediction estimate of predInterSearch ---------------- */

/* TComPrediction::xPredInterLumaBlk (TComPrediction.cpp:405-438): the PU predicted from `ref`
 * at quarter-pel MV (qx, qy) into buf (pitch 64) */
static void FN(mc_luma)(const PIXEL *ref_pu, intptr_t stride, int qx, int qy, int w, int h, int depth, PIXEL *buf)
{
    const int xFrac = qx & 3, yFrac = qy & 3;
    const PIXEL *src = ref_pu + (qx >> 2) + (qy >> 2) * stride;
    if ((xFrac | yFrac) == 0)
        for (int y = 0; y < h; y++) memcpy(buf + 64 * y, src + y * stride, w * sizeof(PIXEL));
    else if (yFrac == 0)
        FN(luma_hpp)(src, stride, buf, 64, xFrac, w, h, depth);
    else if (xFrac == 0)
        FN(luma_vpp)(src, stride, buf, 64, yFrac, w, h, depth);
    else
    {
        int16_t immed[64 * (64 + 8)];
        FN(luma_hps)(src, stride, immed, w, xFrac, 1, w, h, depth);
        FN(luma_vsp)(immed + 3 * w, w, buf, 64, yFrac, w, h, depth);
    }
}

/* TEncSearch.cpp:4346-4368: SATD of the PU against the average of its two predictions, and
 * against the average of the two co-located reference blocks */
static void FN(bipred_cost)(const PIXEL *fenc_pu, intptr_t fstride, const PIXEL *ref0_pu, const PIXEL *ref1_pu, intptr_t rstride,
                            int w, int h, int depth, int mv0x, int mv0y, int mv1x, int mv1y, int32_t out[2])
{
    PIXEL p0[64 * 64], p1[64 * 64], avg[64 * 64];
    FN(mc_luma)(ref0_pu, rstride, mv0x, mv0y, w, h, depth, p0);
    FN(mc_luma)(ref1_pu, rstride, mv1x, mv1y, w, h, depth, p1);
    FN(pixelavg)(avg, 64, p0, 64, p1, 64, w, h);
    out[0] = FN(satd)(fenc_pu, fstride, avg, 64, w, h);
    FN(pixelavg)(avg, 64, ref0_pu, rstride, ref1_pu, rstride, w, h);
    out[1] = FN(satd)(fenc_pu, fstride, avg, 64, w, h);
}


/* ---------------- merge-candidate estimate (compress.cpp:248-271) ---------------- */

/* TComPrediction::xPredInterLumaBlk into 14-bit intermediates (TComPrediction.cpp:441-476):
 * luma_p2s (ipfilter.cpp:155-172), luma_hps, filterVertical_ps_c (:122-153) or luma_hps with row
 * extension + filterVertical_ss_c (:84-119); buf pitch 64 */
static void FN(mc_short)(const PIXEL *ref_pu, intptr_t stride, int qx, int qy, int w, int h, int depth, int16_t *buf)
{
    const int xFrac = qx & 3, yFrac = qy & 3, headRoom = 14 - depth;
    const int shift = 6 - headRoom, offset = -(8192 << shift);
    const PIXEL *src = ref_pu + (qx >> 2) + (qy >> 2) * stride;
    if ((xFrac | yFrac) == 0)
    {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) buf[y * 64 + x] = (int16_t)((int16_t)(src[y * stride + x] << headRoom) - (int16_t)8192);
    }
    else if (yFrac == 0)
        FN(luma_hps)(src, stride, buf, 64, xFrac, 0, w, h, depth);
    else if (xFrac == 0)
    {
        const int16_t *c = orc_luma_filter[yFrac];
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                int sum = 0;
                for (int k = 0; k < 8; k++) sum += (int)src[(y - 3 + k) * stride + x] * c[k];
                buf[y * 64 + x] = (int16_t)((sum + offset) >> shift);
            }
    }
    else
    {
        int16_t immed[64 * (64 + 8)];
        const int16_t *c = orc_luma_filter[yFrac];
        FN(luma_hps)(src, stride, immed, w, xFrac, 1, w, h, depth);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                int sum = 0;
                for (int k = 0; k < 8; k++) sum += (int)immed[(y + k) * w + x] * c[k];
                buf[y * 64 + x] = (int16_t)(sum >> 6);
            }
    }
}

/* Luma motion compensation of one merge candidate of a size x size CU and its sa8d against the
 * source: dir 1 / 2 = one list through the pixel path (xPredInterUni), 3 = both lists through
 * the 14-bit path and TComYuv::addAvg (TComYuv.cpp:573-612) */
static int FN(merge_sa8d)(const PIXEL *fenc_pu, intptr_t fstride, const PIXEL *ref0_pu, intptr_t r0stride, const PIXEL *ref1_pu,
                          intptr_t r1stride, int size, int depth, int dir, int mv0x, int mv0y, int mv1x, int mv1y)
{
    PIXEL pred[64 * 64];
    if (dir == 1) FN(mc_luma)(ref0_pu, r0stride, mv0x, mv0y, size, size, depth, pred);
    else if (dir == 2) FN(mc_luma)(ref1_pu, r1stride, mv1x, mv1y, size, size, depth, pred);
    else
    {
        int16_t s0[64 * 64], s1[64 * 64];
        FN(mc_short)(ref0_pu, r0stride, mv0x, mv0y, size, size, depth, s0);
        FN(mc_short)(ref1_pu, r1stride, mv1x, mv1y, size, size, depth, s1);
        const int shiftNum = 15 - depth, offset = (1 << (shiftNum - 1)) + 2 * 8192, maxv = (1 << depth) - 1;
        for (int y = 0; y < size; y++)
            for (int x = 0; x < size; x++)
            {
                const int v = (s0[y * 64 + x] + s1[y * 64 + x] + offset) >> shiftNum;
                pred[y * 64 + x] = (PIXEL)(v < 0 ? 0 : (v > maxv ? maxv : v));
            }
    }
    return FN(sa8d_inter)(fenc_pu, fstride, pred, 64, size, size);
}


/* ---------------- motion compensation into prediction samples, luma + 4:2:0 chroma ---------------- */

/* The 4-tap chroma chain of TComPrediction::xPredInterChromaBlk, pixel flavour (TComPrediction.cpp:
 * 491-540): copy_pp, chroma filter_hpp / filter_vpp (ipfilter.cpp:194-233, :280-319 with N = 4), or
 * filter_hps with row extension + chroma_vsp (ipfilter.cpp:235-278, :40-81). (ex, ey) is the MV in
 * eighth-pel chroma units (= the luma quarter-pel MV); cw x ch the chroma block; dst pitch ds */
static void FN(mc_chroma)(const PIXEL *ref_blk, intptr_t stride, int ex, int ey, int cw, int ch, int depth, PIXEL *dst, intptr_t ds)
{
    const int xFrac = ex & 7, yFrac = ey & 7, maxv = (1 << depth) - 1, headRoom = 14 - depth;
    const PIXEL *src = ref_blk + (ex >> 3) + (ey >> 3) * stride;
    if ((xFrac | yFrac) == 0)
    {
        for (int y = 0; y < ch; y++) memcpy(dst + ds * y, src + y * stride, cw * sizeof(PIXEL));
        return;
    }
    if (yFrac == 0 || xFrac == 0)
    {
        const int16_t *c = orc_chroma_filter[yFrac == 0 ? xFrac : yFrac];
        const intptr_t step = yFrac == 0 ? 1 : stride;
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)src[y * stride + x + (k - 1) * step] * c[k];
                int16_t val = (int16_t)((sum + 32) >> 6);
                if (val < 0) val = 0;
                if (val > maxv) val = (int16_t)maxv;
                dst[y * ds + x] = (PIXEL)val;
            }
        return;
    }
    int16_t immed[32 * (32 + 3)];
    const int16_t *cx = orc_chroma_filter[xFrac], *cy = orc_chroma_filter[yFrac];
    const int shiftH = 6 - headRoom, offH = -(8192 << shiftH);
    const int shiftV = 6 + headRoom, offV = (1 << (shiftV - 1)) + (8192 << 6);
    for (int y = 0; y < ch + 3; y++)
        for (int x = 0; x < cw; x++)
        {
            int sum = 0;
            for (int k = 0; k < 4; k++) sum += (int)src[(y - 1) * stride + x - 1 + k] * cx[k];
            immed[y * cw + x] = (int16_t)((sum + offH) >> shiftH);
        }
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++)
        {
            int sum = 0;
            for (int k = 0; k < 4; k++) sum += (int)immed[(y + k) * cw + x] * cy[k];
            int16_t val = (int16_t)((sum + offV) >> shiftV);
            if (val < 0) val = 0;
            if (val > maxv) val = (int16_t)maxv;
            dst[y * ds + x] = (PIXEL)val;
        }
}

/* xPredInterChromaBlk into 14-bit intermediates (TComPrediction.cpp:543-593): chroma_p2s
 * (ipfilter.cpp:155-172), filter_hps, filterVertical_ps_c<4> (:118-153), or filter_hps with row
 * extension + filterVertical_ss_c<4> (:83-116); buf pitch 32 */
static void FN(mc_chroma_short)(const PIXEL *ref_blk, intptr_t stride, int ex, int ey, int cw, int ch, int depth, int16_t *buf)
{
    const int xFrac = ex & 7, yFrac = ey & 7, headRoom = 14 - depth;
    const int shift = 6 - headRoom, offset = -(8192 << shift);
    const PIXEL *src = ref_blk + (ex >> 3) + (ey >> 3) * stride;
    if ((xFrac | yFrac) == 0)
    {
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++) buf[y * 32 + x] = (int16_t)((int16_t)(src[y * stride + x] << headRoom) - (int16_t)8192);
    }
    else if (yFrac == 0 || xFrac == 0)
    {
        const int16_t *c = orc_chroma_filter[yFrac == 0 ? xFrac : yFrac];
        const intptr_t step = yFrac == 0 ? 1 : stride;
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)src[y * stride + x + (k - 1) * step] * c[k];
                buf[y * 32 + x] = (int16_t)((sum + offset) >> shift);
            }
    }
    else
    {
        int16_t immed[32 * (32 + 3)];
        const int16_t *cx = orc_chroma_filter[xFrac], *cy = orc_chroma_filter[yFrac];
        for (int y = 0; y < ch + 3; y++)
            for (int x = 0; x < cw; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)src[(y - 1) * stride + x - 1 + k] * cx[k];
                immed[y * cw + x] = (int16_t)((sum + offset) >> shift);
            }
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++) sum += (int)immed[(y + k) * cw + x] * cy[k];
                buf[y * 32 + x] = (int16_t)(sum >> 6);
            }
    }
}

/* TComYuv::addAvg (TComYuv.cpp:573-650): rounded, clipped average of two 14-bit blocks */
static void FN(add_avg)(const int16_t *s0, const int16_t *s1, int pitch, int w, int h, int depth, PIXEL *dst, intptr_t ds)
{
    const int shiftNum = 15 - depth, offset = (1 << (shiftNum - 1)) + 2 * 8192, maxv = (1 << depth) - 1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const int v = (s0[y * pitch + x] + s1[y * pitch + x] + offset) >> shiftNum;
            dst[y * ds + x] = (PIXEL)(v < 0 ? 0 : (v > maxv ? maxv : v));
        }
}

/* TComWeightPrediction::addWeightUni on a 14-bit block (TComWeightPrediction.cpp:433-485 with weight_sp_c,
 * pixel.cpp:494-510) and addWeightBi (:203-313, :46-54); the parameters are getWpScaling's (:534-560) */
static void FN(add_weight)(const int16_t *s0, const int16_t *s1, int pitch, int w, int h, int depth, const orc_mc_weight *wp0,
                           const orc_mc_weight *wp1, int comp, PIXEL *dst, intptr_t ds)
{
    const int maxv = (1 << depth) - 1, shiftNum = 14 - depth;
    if (!s1)
    {
        const int w0 = wp0->w[comp], offset = wp0->offset[comp] * (1 << (depth - 8)), shift = wp0->log2denom[comp] + shiftNum;
        const int round = shift ? (1 << (shift - 1)) : 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                const int v = ((w0 * (s0[y * pitch + x] + 8192) + round) >> shift) + offset;
                dst[y * ds + x] = (PIXEL)(v < 0 ? 0 : (v > maxv ? maxv : v));
            }
        return;
    }
    const int w0 = wp0->w[comp], w1 = wp1->w[comp];
    const int offset = (wp0->offset[comp] + wp1->offset[comp]) * (1 << (depth - 8));
    const int shift = wp0->log2denom[comp] + 1 + shiftNum, round = 1 << (shift - 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const int v = (w0 * (s0[y * pitch + x] + 8192) + w1 * (s1[y * pitch + x] + 8192) + round + offset * (1 << (shift - 1))) >> shift;
            dst[y * ds + x] = (PIXEL)(v < 0 ? 0 : (v > maxv ? maxv : v));
        }
}

/* TComPrediction::motionCompensation of one PU (TComPrediction.cpp:201-286). Without weights: dir 1 / 2 =
 * xPredInterUni into the prediction picture, 3 = xPredInterBi's first branch (both lists into TShortYUV,
 * then addAvg). With weights (flags bit 2; getUseWP / getWPBiPred): every used list goes through the
 * 14-bit flavour, then xWeightedPredictionUni (:212-226) or xWeightedPredictionBi (:334-356).
 * planes: [0] luma, [1] Cb, [2] Cr of the two references
 * and of the prediction picture, all as origins; the PU sits at (px, py) luma samples. */
static void FN(mc_pu)(const PIXEL *const r0[3], const intptr_t r0s[2], const PIXEL *const r1[3], const intptr_t r1s[2],
                      PIXEL *const pred[3], const intptr_t ps[2], int px, int py, int w, int h, int depth, int dir, int flags,
                      int mv0x, int mv0y, int mv1x, int mv1y, const orc_mc_weight *wp0, const orc_mc_weight *wp1)
{
    const int cw = w >> 1, ch = h >> 1, cx = px >> 1, cy = py >> 1;
    const int weighted = flags & 4;
    if (flags & 1)
    {
        PIXEL *d = pred[0] + (intptr_t)py * ps[0] + px;
        if (dir == 3 || weighted)
        {
            int16_t s0[64 * 64], s1[64 * 64];
            if (dir & 1) FN(mc_short)(r0[0] + (intptr_t)py * r0s[0] + px, r0s[0], mv0x, mv0y, w, h, depth, s0);
            if (dir & 2) FN(mc_short)(r1[0] + (intptr_t)py * r1s[0] + px, r1s[0], mv1x, mv1y, w, h, depth, s1);
            if (!weighted) FN(add_avg)(s0, s1, 64, w, h, depth, d, ps[0]);
            else if (dir == 3) FN(add_weight)(s0, s1, 64, w, h, depth, wp0, wp1, 0, d, ps[0]);
            else if (dir == 1) FN(add_weight)(s0, 0, 64, w, h, depth, wp0, 0, 0, d, ps[0]);
            else FN(add_weight)(s1, 0, 64, w, h, depth, wp1, 0, 0, d, ps[0]);
        }
        else
        {
            PIXEL tmp[64 * 64];
            if (dir == 1) FN(mc_luma)(r0[0] + (intptr_t)py * r0s[0] + px, r0s[0], mv0x, mv0y, w, h, depth, tmp);
            else FN(mc_luma)(r1[0] + (intptr_t)py * r1s[0] + px, r1s[0], mv1x, mv1y, w, h, depth, tmp);
            for (int y = 0; y < h; y++) memcpy(d + y * ps[0], tmp + 64 * y, w * sizeof(PIXEL));
        }
    }
    if (flags & 2)
        for (int c = 1; c < 3; c++)
        {
            PIXEL *d = pred[c] + (intptr_t)cy * ps[1] + cx;
            if (dir == 3 || weighted)
            {
                int16_t s0[32 * 32], s1[32 * 32];
                if (dir & 1) FN(mc_chroma_short)(r0[c] + (intptr_t)cy * r0s[1] + cx, r0s[1], mv0x, mv0y, cw, ch, depth, s0);
                if (dir & 2) FN(mc_chroma_short)(r1[c] + (intptr_t)cy * r1s[1] + cx, r1s[1], mv1x, mv1y, cw, ch, depth, s1);
                if (!weighted) FN(add_avg)(s0, s1, 32, cw, ch, depth, d, ps[1]);
                else if (dir == 3) FN(add_weight)(s0, s1, 32, cw, ch, depth, wp0, wp1, c, d, ps[1]);
                else if (dir == 1) FN(add_weight)(s0, 0, 32, cw, ch, depth, wp0, 0, c, d, ps[1]);
                else FN(add_weight)(s1, 0, 32, cw, ch, depth, wp1, 0, c, d, ps[1]);
            }
            else if (dir == 1) FN(mc_chroma)(r0[c] + (intptr_t)cy * r0s[1] + cx, r0s[1], mv0x, mv0y, cw, ch, depth, d, ps[1]);
            else FN(mc_chroma)(r1[c] + (intptr_t)cy * r1s[1] + cx, r1s[1], mv1x, mv1y, cw, ch, depth, d, ps[1]);
        }
}
