// Per-Gaussian and per-(pixel,Gaussian) arithmetic of the surfel rasterizer.
//
// Host+device inline functions: the HIP kernels (preprocess.hip, render.hip) call
// them per lane; tests/host_emu compiles the same header with g++ so the
// hand-derived backward can be checked against autograd of the oracle without a GPU.
//
// Replaces (functionally) the per-Gaussian / per-pixel bodies of the un-vendored
// CUDA extension imported at /root/reference/utils/operations.py:22-25; semantics
// are decisions D1..D12 of oracle/surfel_oracle.py.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AGS_HD __host__ __device__ __forceinline__
#else
#define AGS_HD inline
#endif

#define AGS_TILE 16
#define AGS_NEAR_CULL 0.2f
#define AGS_LOWPASS 0.3f
#define AGS_ALPHA_MAX 0.99f
#define AGS_ALPHA_MIN (1.0f / 255.0f)
#define AGS_T_EPS 1e-4f
#define AGS_COS_MIN 0.02f
#define AGS_DEPTH_A_EPS 1e-6f
#define AGS_FRUSTUM_CLAMP 1.3f

// One projected surfel, 64 B, gathered by the blend kernels.
struct AgsGeom {
    float mx, my, ca, cb;   // pixel-space mean, conic a,b
    float cc, o, dc, gx;    // conic c, log2(opacity), centre depth, depth slope x
    float gy, r, g, b;      // depth slope y, colour
    float nx, ny, nz, conf; // view-space normal (camera facing), confidence
};

// Record accumulated by the blend backward, 64 B.  With gp = alpha * dL/dalpha (0 where the 0.99
// clamp is active) and d = pixel - mean, the first six fields are the RAW MOMENTS of gp over the
// surfel's pixels: everything the conic, the 2-D mean and the opacity need is linear in them, so the
// per-(pixel, surfel) work is 8 multiply-adds and the conversion (ags_preprocess_bwd) runs once per surfel:
//   dL/dconic = (-m2xx/2, -m2xy, -m2yy/2),  dL/do = m0 / o,
//   dL/dmean2D = -(gx*ddc - ca*m1x - cb*m1y,  gy*ddc - cc*m1y - cb*m1x)      (ddc = sum w * dL/ddepth)
struct AgsGeomGrad {
    float m1x, m1y, m2xx, m2xy;
    float m2yy, m0, ddc, dgx;
    float dgy, dr, dg, db;
    float dnx, dny, dnz, pad;
};

struct AgsFrame {
    int H, W, tiles_x, tiles_y;
    float tanfovx, tanfovy, fx, fy;
    float scale_mod;
    int perpix_depth, front_only;
    // AgsCamera.config (device, 5 floats) or nullptr: when set, the kernels take the flags from config[1..4] on the
    // device (ags_frame_flags, ags_internal.h) and the two ints above are only what the host knew
    const float* cfg;
};

AGS_HD float ags_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(x); // v_rcp_f32, 1 ulp
    // opaque to the optimiser: otherwise a later `-(..) * r` is rewritten as a SECOND quarter-rate
    // v_rcp_f32 of the negated argument instead of a free neg modifier on the multiply
    asm("" : "+v"(r));
    return r;
#else
    return 1.0f / x;
#endif
}

AGS_HD float ags_log2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __log2f(x);
#else
    return log2f(x);
#endif
}
AGS_HD float ags_exp2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x); // v_exp_f32
#else
    return exp2f(x);
#endif
}

AGS_HD float ags_affine(const float* M, int j, float x, float y, float z) {
    // [x y z 1] * M, column j, as the fmaf chain the oracle reproduces bit for bit
    return fmaf(x, M[0 * 4 + j], fmaf(y, M[1 * 4 + j], fmaf(z, M[2 * 4 + j], M[3 * 4 + j])));
}

AGS_HD void ags_quat_to_R(const float q[4], float R[9]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Intermediates shared by forward and backward of the per-Gaussian stage.
struct AgsProj {
    float tx, ty, tz;
    float phx, phy, pw;
    float R[9];
    float s[3];
    float J00, J02, J11, J12;
    float rxc, ryc;
    int clx, cly; // 1 when x/z (y/z) was clamped
    float Tm[6];  // J*A, 2x3
    float U[6];   // Tm*R*S, 2x3
    float a, b, c, det;
    float nv[3], sgn, nc, len, ncc;
    int ncc_clamped;
};

AGS_HD bool ags_project(const AgsFrame& F, const float* V, const float* P, const float p[3],
                        const float sc[3], const float q[4], AgsProj& w) {
    w.tx = ags_affine(V, 0, p[0], p[1], p[2]);
    w.ty = ags_affine(V, 1, p[0], p[1], p[2]);
    w.tz = ags_affine(V, 2, p[0], p[1], p[2]);
    if (!(w.tz > AGS_NEAR_CULL)) return false;
    w.phx = ags_affine(P, 0, p[0], p[1], p[2]);
    w.phy = ags_affine(P, 1, p[0], p[1], p[2]);
    const float phw = ags_affine(P, 3, p[0], p[1], p[2]);
    w.pw = 1.0f / (phw + 1e-7f);
    ags_quat_to_R(q, w.R);
    for (int k = 0; k < 3; ++k) w.s[k] = sc[k] * F.scale_mod;
    const float limx = AGS_FRUSTUM_CLAMP * F.tanfovx, limy = AGS_FRUSTUM_CLAMP * F.tanfovy;
    const float iz = 1.0f / w.tz;
    const float rx = w.tx * iz, ry = w.ty * iz;
    w.rxc = fminf(limx, fmaxf(-limx, rx));
    w.ryc = fminf(limy, fmaxf(-limy, ry));
    w.clx = (rx != w.rxc);
    w.cly = (ry != w.ryc);
    w.J00 = F.fx * iz;
    w.J02 = -F.fx * w.rxc * iz;
    w.J11 = F.fy * iz;
    w.J12 = -F.fy * w.ryc * iz;
    // A[j][k] = V[k*4+j]  (t = A p + b)
    for (int k = 0; k < 3; ++k) {
        w.Tm[k] = w.J00 * V[k * 4 + 0] + w.J02 * V[k * 4 + 2];
        w.Tm[3 + k] = w.J11 * V[k * 4 + 1] + w.J12 * V[k * 4 + 2];
    }
    for (int k = 0; k < 3; ++k) {
        w.U[k] = w.s[k] * (w.Tm[0] * w.R[k] + w.Tm[1] * w.R[3 + k] + w.Tm[2] * w.R[6 + k]);
        w.U[3 + k] = w.s[k] * (w.Tm[3] * w.R[k] + w.Tm[4] * w.R[3 + k] + w.Tm[5] * w.R[6 + k]);
    }
    w.a = w.U[0] * w.U[0] + w.U[1] * w.U[1] + w.U[2] * w.U[2] + AGS_LOWPASS;
    w.b = w.U[0] * w.U[3] + w.U[1] * w.U[4] + w.U[2] * w.U[5];
    w.c = w.U[3] * w.U[3] + w.U[4] * w.U[4] + w.U[5] * w.U[5] + AGS_LOWPASS;
    w.det = w.a * w.c - w.b * w.b;
    if (!(w.det > 0.f)) return false;
    // view-space normal = A * R[:,2], flipped towards the camera (D11)
    for (int j = 0; j < 3; ++j)
        w.nv[j] = V[0 * 4 + j] * w.R[2] + V[1 * 4 + j] * w.R[5] + V[2 * 4 + j] * w.R[8];
    const float d = w.nv[0] * w.tx + w.nv[1] * w.ty + w.nv[2] * w.tz;
    w.sgn = (d > 0.f) ? -1.f : 1.f;
    if (F.front_only && d > 0.f) return false; // D10
    w.nc = w.sgn * d;
    w.len = sqrtf(w.tx * w.tx + w.ty * w.ty + w.tz * w.tz);
    const float lim = -AGS_COS_MIN * w.len;
    w.ncc_clamped = !(w.nc <= lim);
    w.ncc = w.ncc_clamped ? lim : w.nc;
    return true;
}

// Forward of the per-Gaussian stage (F1).  Returns false when culled.
AGS_HD bool ags_preprocess_fwd(const AgsFrame& F, const float* V, const float* P, const float p[3],
                               const float sc[3], const float q[4], float opacity, const float col[3],
                               float conf, float m2dx, float m2dy, AgsGeom& g, int& radius, int rect[4]) {
    AgsProj w;
    if (!ags_project(F, V, P, p, sc, q, w)) return false;
    const float idet = 1.0f / w.det;
    const float mid = 0.5f * (w.a + w.c);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - w.det));
    const float rad = ceilf(3.0f * sqrtf(lam));
    const float mx = ((w.phx * w.pw + 1.0f) * F.W - 1.0f) * 0.5f + m2dx;
    const float my = ((w.phy * w.pw + 1.0f) * F.H - 1.0f) * 0.5f + m2dy;
    // D3: C-style truncation then clamp
    int x0 = (int)((mx - rad) / AGS_TILE), x1 = (int)((mx + rad + AGS_TILE - 1) / AGS_TILE);
    int y0 = (int)((my - rad) / AGS_TILE), y1 = (int)((my + rad + AGS_TILE - 1) / AGS_TILE);
    x0 = x0 < 0 ? 0 : (x0 > F.tiles_x ? F.tiles_x : x0);
    x1 = x1 < 0 ? 0 : (x1 > F.tiles_x ? F.tiles_x : x1);
    y0 = y0 < 0 ? 0 : (y0 > F.tiles_y ? F.tiles_y : y0);
    y1 = y1 < 0 ? 0 : (y1 > F.tiles_y ? F.tiles_y : y1);
    if ((x1 - x0) * (y1 - y0) <= 0) return false;
    rect[0] = x0; rect[1] = y0; rect[2] = x1; rect[3] = y1;
    radius = (int)rad;
    g.mx = mx; g.my = my;
    g.ca = w.c * idet; g.cb = -w.b * idet; g.cc = w.a * idet;
    g.o = ags_log2(opacity); g.dc = w.tz; // log2: alpha = exp2(power*log2(e) + g.o), one multiply less per pixel
    if (F.perpix_depth) {
        const float qq = -(w.tz * w.tz) / w.ncc;
        g.gx = qq * (w.sgn * w.nv[0]) / F.fx;
        g.gy = qq * (w.sgn * w.nv[1]) / F.fy;
    } else {
        g.gx = 0.f; g.gy = 0.f;
    }
    g.r = col[0]; g.g = col[1]; g.b = col[2];
    g.nx = w.sgn * w.nv[0]; g.ny = w.sgn * w.nv[1]; g.nz = w.sgn * w.nv[2];
    g.conf = conf;
    return true;
}

// Backward of the per-Gaussian stage (B2+B3 fused).  `dg` is what the blend
// backward accumulated for this Gaussian.  Outputs are written (not accumulated).
AGS_HD void ags_preprocess_bwd(const AgsFrame& F, const float* V, const float* P, const float p[3],
                               const float sc[3], const float q[4], float opacity, const AgsGeomGrad& dg,
                               float dmean[3], float dscale[3], float dquat[4], float* dopacity,
                               float dcolor[3], float dmean2d[2]) {
    AgsProj w;
    // caller guarantees the Gaussian was visible in the forward pass
    ags_project(F, V, P, p, sc, q, w);
    // the blend backward accumulates sum(alpha * dL/dalpha) = o * sum(G * dL/dalpha)
    *dopacity = opacity > 0.f ? dg.m0 / opacity : 0.f;
    dcolor[0] = dg.dr; dcolor[1] = dg.dg; dcolor[2] = dg.db;

    // ---- raw moments -> gradients of the conic and of the 2-D mean (see AgsGeomGrad)
    const float idet = 1.0f / w.det, idet2 = idet * idet;
    const float a = w.a, b = w.b, c = w.c;
    const float ca = c * idet, cb = -b * idet, cc = a * idet;
    float gx = 0.f, gy = 0.f;
    if (F.perpix_depth) { // the depth slopes of the forward record (ags_preprocess_fwd)
        const float qq0 = -(w.tz * w.tz) / w.ncc;
        gx = qq0 * (w.sgn * w.nv[0]) / F.fx;
        gy = qq0 * (w.sgn * w.nv[1]) / F.fy;
    }
    const float dmx = -(gx * dg.ddc - ca * dg.m1x - cb * dg.m1y); // d = pixel - mean
    const float dmy = -(gy * dg.ddc - cc * dg.m1y - cb * dg.m1x);
    dmean2d[0] = dmx; dmean2d[1] = dmy;

    // ---- conic -> cov2D (a,b,c)
    const float dka = -0.5f * dg.m2xx, dkb = -dg.m2xy, dkc = -0.5f * dg.m2yy; // grads of conic (c/det, -b/det, a/det)
    const float da = dka * (-c * c * idet2) + dkb * (b * c * idet2) + dkc * (idet - a * c * idet2);
    const float db = dka * (2.f * b * c * idet2) + dkb * (-idet - 2.f * b * b * idet2) + dkc * (2.f * a * b * idet2);
    const float dc = dka * (idet - a * c * idet2) + dkb * (a * b * idet2) + dkc * (-a * a * idet2);
    // ---- cov2D -> U (2x3)
    float dU[6];
    for (int k = 0; k < 3; ++k) {
        dU[k] = 2.f * da * w.U[k] + db * w.U[3 + k];
        dU[3 + k] = 2.f * dc * w.U[3 + k] + db * w.U[k];
    }
    // U[r][k] = s[k] * sum_j Tm[r][j] R[j][k]
    float dTm[6] = {0, 0, 0, 0, 0, 0};
    float dR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float ds[3];
    for (int k = 0; k < 3; ++k) {
        const float tr0 = w.Tm[0] * w.R[k] + w.Tm[1] * w.R[3 + k] + w.Tm[2] * w.R[6 + k];
        const float tr1 = w.Tm[3] * w.R[k] + w.Tm[4] * w.R[3 + k] + w.Tm[5] * w.R[6 + k];
        ds[k] = dU[k] * tr0 + dU[3 + k] * tr1;
        for (int j = 0; j < 3; ++j) {
            dTm[j] += dU[k] * w.s[k] * w.R[j * 3 + k];
            dTm[3 + j] += dU[3 + k] * w.s[k] * w.R[j * 3 + k];
            dR[j * 3 + k] += w.s[k] * (dU[k] * w.Tm[j] + dU[3 + k] * w.Tm[3 + j]);
        }
    }
    for (int k = 0; k < 3; ++k) dscale[k] = ds[k] * F.scale_mod;
    // Tm[0][k] = J00*A[0][k] + J02*A[2][k]; Tm[1][k] = J11*A[1][k] + J12*A[2][k]; A[j][k] = V[k*4+j]
    float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
    for (int k = 0; k < 3; ++k) {
        dJ00 += dTm[k] * V[k * 4 + 0];
        dJ02 += dTm[k] * V[k * 4 + 2];
        dJ11 += dTm[3 + k] * V[k * 4 + 1];
        dJ12 += dTm[3 + k] * V[k * 4 + 2];
    }
    const float iz = 1.0f / w.tz, iz2 = iz * iz;
    float dtx = 0.f, dty = 0.f, dtz = 0.f;
    // J00 = fx/tz ; J02 = -fx*rxc/tz, rxc = clamp(tx/tz)
    dtz += dJ00 * (-F.fx * iz2) + dJ11 * (-F.fy * iz2);
    dtz += dJ02 * (F.fx * w.rxc * iz2) + dJ12 * (F.fy * w.ryc * iz2);
    if (!w.clx) { // d rxc / d tx = 1/tz ; d rxc / d tz = -tx/tz^2
        const float drx = dJ02 * (-F.fx * iz);
        dtx += drx * iz;
        dtz += drx * (-w.tx * iz2);
    }
    if (!w.cly) {
        const float dry = dJ12 * (-F.fy * iz);
        dty += dry * iz;
        dtz += dry * (-w.ty * iz2);
    }
    // ---- depth centre, slopes, normal
    dtz += dg.ddc;
    float dn[3] = {dg.dnx, dg.dny, dg.dnz}; // wrt flipped normal n = sgn*nv
    if (F.perpix_depth) {
        const float nx = w.sgn * w.nv[0], ny = w.sgn * w.nv[1];
        const float qq = -(w.tz * w.tz) / w.ncc;
        dn[0] += dg.dgx * qq / F.fx;
        dn[1] += dg.dgy * qq / F.fy;
        const float dqq = dg.dgx * nx / F.fx + dg.dgy * ny / F.fy;
        dtz += dqq * (-2.f * w.tz / w.ncc);
        const float dncc = dqq * (w.tz * w.tz) / (w.ncc * w.ncc);
        if (w.ncc_clamped) {
            const float k = dncc * (-AGS_COS_MIN) / w.len;
            dtx += k * w.tx; dty += k * w.ty; dtz += k * w.tz;
        } else {
            // nc = n . t
            dn[0] += dncc * w.tx; dn[1] += dncc * w.ty; dn[2] += dncc * w.tz;
            dtx += dncc * nx; dty += dncc * ny; dtz += dncc * (w.sgn * w.nv[2]);
        }
    }
    // n = sgn * A * R[:,2]  ->  dR[:,2] += sgn * A^T dn ; (A^T dn)_k = sum_j A[j][k] dn_j = sum_j V[k*4+j] dn_j
    for (int k = 0; k < 3; ++k)
        dR[k * 3 + 2] += w.sgn * (V[k * 4 + 0] * dn[0] + V[k * 4 + 1] * dn[1] + V[k * 4 + 2] * dn[2]);
    // ---- mean2D -> clip space
    const float dphx = dmx * (0.5f * F.W) * w.pw;
    const float dphy = dmy * (0.5f * F.H) * w.pw;
    const float dpw = dmx * (0.5f * F.W) * w.phx + dmy * (0.5f * F.H) * w.phy;
    const float dphw = -w.pw * w.pw * dpw;
    for (int i = 0; i < 3; ++i) {
        dmean[i] = P[i * 4 + 0] * dphx + P[i * 4 + 1] * dphy + P[i * 4 + 3] * dphw
                 + V[i * 4 + 0] * dtx + V[i * 4 + 1] * dty + V[i * 4 + 2] * dtz;
    }
    // ---- R -> quaternion (w,x,y,z)
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    dquat[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    dquat[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
    dquat[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    dquat[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}

// ---------------------------------------------------------------------------------
// Conservative reach test used to skip work, never to change results: can ANY point of the
// box [x0,x1]x[y0,y1] (pixel-centre extents) have alpha >= 1/255 for this surfel?
// alpha >= 1/255  <=>  q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 <= 2 ln(255 o).  The minimum of
// the convex q over the box is 0 when the centre is inside, else it lies on a face nearest
// to the centre.  A small margin absorbs rounding differences with the per-pixel test.
AGS_HD bool ags_reaches_box(const AgsGeom& g, float x0, float x1, float y0, float y1) {
    // g.o = log2(opacity): 2 ln(255 o) = 2 ln2 (log2 255 + g.o)
    const float l255o = 7.99435344f + g.o;
    if (!(l255o >= 0.f)) return false;
    const float tau = (2.f * 0.69314718f * 1.0001f) * l255o + 1e-3f;
    const float ax0 = x0 - g.mx, ax1 = x1 - g.mx, ay0 = y0 - g.my, ay1 = y1 - g.my;
    const bool inx = (ax0 <= 0.f) && (ax1 >= 0.f), iny = (ay0 <= 0.f) && (ay1 >= 0.f);
    if (inx && iny) return true;
    float best = 3.0e38f;
    if (!inx) {
        const float dx = ax0 > 0.f ? ax0 : ax1;
        const float dy = fminf(fmaxf(-(g.cb / g.cc) * dx, ay0), ay1);
        best = fminf(best, g.ca * dx * dx + 2.f * g.cb * dx * dy + g.cc * dy * dy);
    }
    if (!iny) {
        const float dy = ay0 > 0.f ? ay0 : ay1;
        const float dx = fminf(fmaxf(-(g.cb / g.ca) * dy, ax0), ax1);
        best = fminf(best, g.ca * dx * dx + 2.f * g.cb * dx * dy + g.cc * dy * dy);
    }
    return best <= tau;
}

// ---------------------------------------------------------------------------------
// Blend step, forward.  Per-pixel accumulator for one of the lane's pixels.
struct AgsPix {
    float T;
    float c0, c1, c2, n0, n1, n2, d, cf;
    uint32_t last;
    int done;
};

AGS_HD void ags_pix_init(AgsPix& s, bool inside) {
    s.T = 1.f; s.c0 = s.c1 = s.c2 = s.n0 = s.n1 = s.n2 = s.d = s.cf = 0.f;
    s.last = 0; s.done = inside ? 0 : 1;
}

// alpha of Gaussian g at pixel (px,py); returns whether the pixel takes it (D4, before
// the transmittance test).
AGS_HD bool ags_alpha(const AgsGeom& g, float px, float py, float& dx, float& dy, float& alpha) {
    dx = px - g.mx; dy = py - g.my;
    const float power = -0.5f * (g.ca * dx * dx + g.cc * dy * dy) - g.cb * dx * dy;
    alpha = fminf(AGS_ALPHA_MAX, ags_exp2(fmaf(power, 1.44269504f, g.o))); // o * exp(power), g.o = log2(o)
    return (power <= 0.f) && (alpha >= AGS_ALPHA_MIN);
}

// ---------------------------------------------------------------------------------
// Quadrant-local polynomial form of a surfel (what the blend kernels stage in LDS).
// With q = pixel - quadrant centre and o = mean - quadrant centre (so d = pixel - mean = q - o) the exponent of
// alpha = min(0.99, opacity * exp(power(d))) is a quadratic in q whose coefficients depend on the surfel and the
// quadrant only:
//   log2(opacity) + log2(e) * power = E0 + qx (E1 + E3 qx + E4 qy) + qy (E2 + E5 qy)          (5 fmas per pixel)
// and the surfel's depth at the pixel is D0 + gx qx + gy qy.  The per-pixel work needs neither the pixel's
// coordinates nor its offset from the mean: 5 instructions fewer per (pixel, surfel) than ags_alpha, and the
// 20-odd instructions of ags_quad_coeffs run once per staged surfel and wave.  D4's "skip power > 0" needs no
// test in this form: the conic is positive definite (D2's low-pass), so power <= 0 up to rounding - and a
// rounding-level positive exponent at a surfel's centre must NOT drop the pixel.
struct AgsQuadShared { float E3, E4, E5; };          // per surfel
struct AgsQuadCoef { float E0, E1, E2, D0; };        // per (surfel, quadrant)
#define AGS_LOG2E 1.44269504f
AGS_HD void ags_quad_shared(const AgsGeom& g, AgsQuadShared& s) {
    s.E3 = (-0.5f * AGS_LOG2E) * g.ca; s.E5 = (-0.5f * AGS_LOG2E) * g.cc; s.E4 = -AGS_LOG2E * g.cb;
}
AGS_HD void ags_quad_coeffs(const AgsGeom& g, float cx, float cy, AgsQuadCoef& q) {
    const float ox = g.mx - cx, oy = g.my - cy;
    const float ax = g.ca * ox + g.cb * oy, ay = g.cc * oy + g.cb * ox;   // -d power / d q at the quadrant centre
    q.E1 = AGS_LOG2E * ax; q.E2 = AGS_LOG2E * ay;
    q.E0 = fmaf(-0.5f * AGS_LOG2E, ax * ox + ay * oy, g.o);               // log2(opacity) + log2(e) * power(centre)
    q.D0 = g.dc - g.gx * ox - g.gy * oy;
}
AGS_HD float ags_alpha_quad(const AgsQuadShared& s, const AgsQuadCoef& q, float qx, float qy) {
    const float t1 = fmaf(s.E4, qy, fmaf(s.E3, qx, q.E1));
    const float t2 = fmaf(s.E5, qy, q.E2);
    return fminf(AGS_ALPHA_MAX, ags_exp2(fmaf(qy, t2, fmaf(qx, t1, q.E0))));
}

// Blend a Gaussian whose alpha test passed; returns weight w (0 when the pixel stops).
// `pos1` = 1-based position in the tile list.
AGS_HD float ags_blend_apply(AgsPix& s, const AgsGeom& g, float dx, float dy, float alpha, uint32_t pos1) {
    // branchless on purpose: per-lane selects, and no conditional stores into `s`.  A lane that
    // does not take the surfel may be passed alpha = 0: nothing changes for it.
    const float testT = s.T * (1.f - alpha);
    const bool stop = testT < AGS_T_EPS;
    const float w = stop ? 0.f : alpha * s.T;
    s.c0 += w * g.r; s.c1 += w * g.g; s.c2 += w * g.b;
    s.n0 += w * g.nx; s.n1 += w * g.ny; s.n2 += w * g.nz;
    s.d += w * (g.dc + g.gx * dx + g.gy * dy);
    s.cf += w * g.conf;
    s.T = stop ? s.T : testT;
    s.last = (stop || !(alpha > 0.f)) ? s.last : pos1;
    s.done = stop ? 1 : s.done;
    return w;
}

// the same with the surfel's payload handed over piecewise (quadrant-local form: dpix = D0 + gx qx + gy qy)
AGS_HD float ags_blend_apply_q(AgsPix& s, float r, float g, float b, float nx, float ny, float nz, float conf,
                               float dpix, float alpha, uint32_t pos1) {
    const float testT = s.T * (1.f - alpha);
    const bool stop = testT < AGS_T_EPS;
    const float w = stop ? 0.f : alpha * s.T;
    s.c0 += w * r; s.c1 += w * g; s.c2 += w * b;
    s.n0 += w * nx; s.n1 += w * ny; s.n2 += w * nz;
    s.d += w * dpix;
    s.cf += w * conf;
    s.T = stop ? s.T : testT;
    s.last = (stop || !(alpha > 0.f)) ? s.last : pos1;
    s.done = stop ? 1 : s.done;
    return w;
}

AGS_HD float ags_blend_fwd(AgsPix& s, const AgsGeom& g, float px, float py, uint32_t pos1) {
    if (s.done) return 0.f;
    float dx, dy, alpha;
    if (!ags_alpha(g, px, py, dx, dy, alpha)) return 0.f;
    return ags_blend_apply(s, g, dx, dy, alpha, pos1);
}

// Per-pixel constants of the backward pass, derived from the incoming image gradients.
struct AgsPixGrad {
    float dC0, dC1, dC2, dN0, dN1, dN2, dDn, dCf, dA; // dL/d(sum w*feature) per channel
    float T;      // running transmittance (starts at T_final)
    float S;      // suffix sum of w*g plus T_final*(dC . bg)
    uint32_t last;
};

AGS_HD void ags_pixgrad_init(AgsPixGrad& s, const float dC[3], const float dN[3], float dDepth, float dOpac,
                             float dConf, float depth_out, float opac_out, float T_final, uint32_t last,
                             const float bg[3], int normalize_depth) {
    s.dC0 = dC[0]; s.dC1 = dC[1]; s.dC2 = dC[2];
    s.dN0 = dN[0]; s.dN1 = dN[1]; s.dN2 = dN[2];
    s.dCf = dConf;
    if (normalize_depth) {
        const float Ae = fmaxf(opac_out, AGS_DEPTH_A_EPS);
        s.dDn = dDepth / Ae;
        s.dA = dOpac - ((opac_out > AGS_DEPTH_A_EPS) ? dDepth * depth_out / Ae : 0.f);
    } else {
        s.dDn = dDepth;
        s.dA = dOpac;
    }
    s.T = T_final;
    s.S = T_final * (dC[0] * bg[0] + dC[1] * bg[1] + dC[2] * bg[2]);
    s.last = last;
}

// One back-to-front step for a Gaussian whose alpha test passed at this pixel
// (and pos1 <= s.last); accumulates the pixel's contribution into `acc`.
AGS_HD void ags_blend_bwd_apply(AgsPixGrad& s, const AgsGeom& g, float dx, float dy, float alpha, AgsGeomGrad& acc) {
    // branchless: a lane that does not take the surfel may be passed alpha = 0 (iom = 1, w = 0,
    // every contribution vanishes, T and S are unchanged)
    const float iom = ags_rcp(1.f - alpha); // one reciprocal serves T/(1-a) and S/(1-a)
    s.T = s.T * iom; // transmittance in front of this Gaussian
    const float w = alpha * s.T;
    const float dpix = g.dc + g.gx * dx + g.gy * dy;
    const float gsum = s.dC0 * g.r + s.dC1 * g.g + s.dC2 * g.b + s.dN0 * g.nx + s.dN1 * g.ny + s.dN2 * g.nz
                     + s.dDn * dpix + s.dCf * g.conf + s.dA;
    const float dalpha = s.T * gsum - s.S * iom;
    s.S += w * gsum;
    acc.dr += w * s.dC0; acc.dg += w * s.dC1; acc.db += w * s.dC2;
    acc.dnx += w * s.dN0; acc.dny += w * s.dN1; acc.dnz += w * s.dN2;
    const float wd = w * s.dDn;
    acc.ddc += wd; acc.dgx += wd * dx; acc.dgy += wd * dy;
    // the 0.99 clamp passes no gradient; gp = dL/dpower = o*G*dalpha = alpha*dalpha when unclamped.
    // m0 collects alpha*dalpha = o * (G*dalpha); the per-Gaussian backward divides by o once.
    const float gp = (alpha < AGS_ALPHA_MAX) ? alpha * dalpha : 0.f;
    const float t = gp * dx, u = gp * dy;
    acc.m0 += gp;
    acc.m1x += t; acc.m1y += u;
    acc.m2xx += t * dx; acc.m2xy += t * dy; acc.m2yy += u * dy;
}

// `pos1` is the Gaussian's 1-based position in the tile list. Returns true if it contributed.
AGS_HD bool ags_blend_bwd(AgsPixGrad& s, const AgsGeom& g, float px, float py, uint32_t pos1, AgsGeomGrad& acc) {
    if (pos1 > s.last) return false;
    float dx, dy, alpha;
    if (!ags_alpha(g, px, py, dx, dy, alpha)) return false;
    ags_blend_bwd_apply(s, g, dx, dy, alpha, acc);
    return true;
}
