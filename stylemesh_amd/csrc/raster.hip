// R1: UV / angle / depth rasteriser - the step immediately before the hot path.
//
// Reference component replaced (lukasHoel/stylemesh): the OpenGL renderer scripts/scannet/render_uv (GLFW + GLEW +
// Assimp + OpenCV; src/renderer/renderer.cpp:165-224, scannet_renderer.cpp:19-84, shader/{uvmap,angle,depth}.*,
// include/util.h:11-35), which draws the UV-parameterised scene mesh once per pose and shader and reads the float
// framebuffer back. Same outputs per pixel: interpolated texture coordinate (u, v) of the nearest surface, cosine
// between the interpolated vertex normal and the direction to the eye (clamped at 0, shader/angle.frag), and the
// view-space depth (what LinearizeDepth(gl_FragCoord.z) of shader/depth.frag recovers); background = 0.
// Camera: the pose's x right / y down / z forward frame (ScanNet), pixel (i, j) sampled at (i + 0.5, j + 0.5) with
// u = fx X/Z + cx (the OpenGL sample position under include/util.h's projection), row 0 = top of the image.
// Parity: UNPINNED against the reference (its renderer needs an OpenGL context and four libraries this image lacks);
// pinned instead on an analytic ray caster (stylemesh_amd/data/synthetic.py BoxRoom) - see DESIGN.md.
//
// Two passes: (1) one thread per triangle: camera transform, near-plane clip (up to two triangles), screen bounding
// box, edge functions at the pixel centres, perspective-correct depth, 64-bit atomicMin of (depth bits << 32 | face);
// (2) one thread per pixel: ray / plane intersection with the winning face in camera space -> barycentrics ->
// attributes (independent of how the face was clipped).
#include <algorithm>

#include "common.h"

namespace sm {

struct RasterCam {
    float r[9];      // world -> camera rotation, row-major
    float t[3];      // camera = r * world + t
    float fx, fy, cx, cy;
    int H, W;
    float znear, zfar;
};

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 cam_point(const RasterCam& c, const float* p) {
    return V3{c.r[0] * p[0] + c.r[1] * p[1] + c.r[2] * p[2] + c.t[0], c.r[3] * p[0] + c.r[4] * p[1] + c.r[5] * p[2] + c.t[1],
              c.r[6] * p[0] + c.r[7] * p[1] + c.r[8] * p[2] + c.t[2]};
}
__device__ __forceinline__ V3 lerp3(V3 a, V3 b, float s) { return V3{a.x + s * (b.x - a.x), a.y + s * (b.y - a.y), a.z + s * (b.z - a.z)}; }

struct BigList {
    float* tris;                 // [cap][10]: camera-space vertices a, b, d (9 floats) + face id (bits)
    int* count;
    int cap;
};

// rasterise pixels [start, start + step, ...) of the triangle's bounding box (row-major box index)
__device__ void raster_tri_pixels(const RasterCam& c, V3 a, V3 b, V3 d, unsigned face, unsigned long long* zbuf, int start,
                                  int step) {
    const float ax = c.fx * a.x / a.z + c.cx, ay = c.fy * a.y / a.z + c.cy;
    const float bx = c.fx * b.x / b.z + c.cx, by = c.fy * b.y / b.z + c.cy;
    const float dx = c.fx * d.x / d.z + c.cx, dy = c.fy * d.y / d.z + c.cy;
    const float area = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax);
    if (area == 0.f || !(fabsf(area) < 3.0e38f)) return;
    const int x0 = max(0, (int)ceilf(fminf(ax, fminf(bx, dx)) - 0.5f)), x1 = min(c.W - 1, (int)floorf(fmaxf(ax, fmaxf(bx, dx)) - 0.5f));
    const int y0 = max(0, (int)ceilf(fminf(ay, fminf(by, dy)) - 0.5f)), y1 = min(c.H - 1, (int)floorf(fmaxf(ay, fmaxf(by, dy)) - 0.5f));
    if (x1 < x0 || y1 < y0) return;
    const int bw = x1 - x0 + 1, n = bw * (y1 - y0 + 1);
    const float inv_area = 1.f / area, iza = 1.f / a.z, izb = 1.f / b.z, izd = 1.f / d.z;
    for (int k = start; k < n; k += step) {
        const int y = y0 + k / bw, x = x0 + k % bw;
        const float px = x + 0.5f, py = y + 0.5f;
        // barycentrics of the pixel centre (either winding: no back-face culling, as the reference)
        const float w0 = ((bx - px) * (dy - py) - (by - py) * (dx - px)) * inv_area;
        const float w1 = ((dx - px) * (ay - py) - (dy - py) * (ax - px)) * inv_area;
        const float w2 = 1.f - w0 - w1;
        if (w0 < 0.f || w1 < 0.f || w2 < 0.f) continue;
        const float z = 1.f / (w0 * iza + w1 * izb + w2 * izd);
        if (!(z >= c.znear && z <= c.zfar)) continue;
        atomicMin(&zbuf[y * c.W + x], ((unsigned long long)__float_as_uint(z) << 32) | face);
    }
}

// small triangles are rasterised by the calling thread; triangles whose screen box exceeds 256 pixels are queued for
// raster_big_kernel (one block each) - a wall that fills the view must not serialise on one lane
__device__ void raster_tri(const RasterCam& c, V3 a, V3 b, V3 d, unsigned face, unsigned long long* zbuf, BigList big) {
    const float ax = c.fx * a.x / a.z + c.cx, ay = c.fy * a.y / a.z + c.cy;
    const float bx = c.fx * b.x / b.z + c.cx, by = c.fy * b.y / b.z + c.cy;
    const float dx = c.fx * d.x / d.z + c.cx, dy = c.fy * d.y / d.z + c.cy;
    const float w = fminf(fmaxf(ax, fmaxf(bx, dx)), (float)c.W) - fmaxf(fminf(ax, fminf(bx, dx)), 0.f);
    const float h = fminf(fmaxf(ay, fmaxf(by, dy)), (float)c.H) - fmaxf(fminf(ay, fminf(by, dy)), 0.f);
    if (w > 0.f && h > 0.f && w * h > 256.f && big.tris) {
        const int slot = atomicAdd(big.count, 1);
        if (slot < big.cap) {
            float* t = big.tris + 10 * (size_t)slot;
            t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = b.x; t[4] = b.y; t[5] = b.z; t[6] = d.x; t[7] = d.y; t[8] = d.z;
            t[9] = __uint_as_float(face);
            return;
        }
    }
    raster_tri_pixels(c, a, b, d, face, zbuf, 0, 1);
}

__global__ __launch_bounds__(256) void raster_big_kernel(RasterCam c, BigList big, unsigned long long* zbuf) {
    const int n = min(*big.count, big.cap);
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const float* t = big.tris + 10 * (size_t)k;
        raster_tri_pixels(c, V3{t[0], t[1], t[2]}, V3{t[3], t[4], t[5]}, V3{t[6], t[7], t[8]}, __float_as_uint(t[9]), zbuf,
                          threadIdx.x, 256);
    }
}

__global__ __launch_bounds__(256) void raster_depth_kernel(RasterCam c, const float* __restrict__ verts,
                                                           const int* __restrict__ faces, int n_faces,
                                                           unsigned long long* zbuf, BigList big) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_faces) return;
    V3 p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = cam_point(c, verts + 3 * (size_t)faces[3 * f + k]);
    const bool in0 = p[0].z >= c.znear, in1 = p[1].z >= c.znear, in2 = p[2].z >= c.znear;
    const int n_in = in0 + in1 + in2;
    if (n_in == 0) return;
    if (n_in == 3) { raster_tri(c, p[0], p[1], p[2], f, zbuf, big); return; }
    // clip against z = znear: rotate so that the vertex pattern is (in, out, out) or (in, in, out)
    V3 a = p[0], b = p[1], d = p[2];
    if (n_in == 1) {
        if (in1) { a = p[1]; b = p[2]; d = p[0]; } else if (in2) { a = p[2]; b = p[0]; d = p[1]; }
        const V3 ab = lerp3(a, b, (c.znear - a.z) / (b.z - a.z)), ad = lerp3(a, d, (c.znear - a.z) / (d.z - a.z));
        raster_tri(c, a, ab, ad, f, zbuf, big);
    } else {
        if (!in0) { a = p[1]; b = p[2]; d = p[0]; } else if (!in1) { a = p[2]; b = p[0]; d = p[1]; }   // d is outside
        const V3 ad = lerp3(a, d, (c.znear - a.z) / (d.z - a.z)), bd = lerp3(b, d, (c.znear - b.z) / (d.z - b.z));
        raster_tri(c, a, b, bd, f, zbuf, big);
        raster_tri(c, a, bd, ad, f, zbuf, big);
    }
}

__global__ __launch_bounds__(256) void raster_shade_kernel(RasterCam c, const float* __restrict__ verts,
                                                           const float* __restrict__ normals, const float* __restrict__ uvs,
                                                           const int* __restrict__ faces,
                                                           const unsigned long long* __restrict__ zbuf, float* __restrict__ uv_out,
                                                           float* __restrict__ angle_out, float* __restrict__ depth_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c.H * c.W) return;
    const unsigned long long key = zbuf[i];
    float u = 0.f, v = 0.f, ang = 0.f, dep = 0.f;
    if (key != ~0ull) {
        const unsigned f = (unsigned)(key & 0xffffffffu);
        const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const V3 a = cam_point(c, verts + 3 * (size_t)i0), b = cam_point(c, verts + 3 * (size_t)i1), d = cam_point(c, verts + 3 * (size_t)i2);
        const int y = i / c.W, x = i - y * c.W;
        const V3 dir{(x + 0.5f - c.cx) / c.fx, (y + 0.5f - c.cy) / c.fy, 1.f};
        // Moeller-Trumbore with the ray origin at the eye
        const V3 e1{b.x - a.x, b.y - a.y, b.z - a.z}, e2{d.x - a.x, d.y - a.y, d.z - a.z};
        const V3 pv{dir.y * e2.z - dir.z * e2.y, dir.z * e2.x - dir.x * e2.z, dir.x * e2.y - dir.y * e2.x};
        const float det = e1.x * pv.x + e1.y * pv.y + e1.z * pv.z;
        const float inv = 1.f / det;
        const V3 tv{-a.x, -a.y, -a.z};
        const float b1 = (tv.x * pv.x + tv.y * pv.y + tv.z * pv.z) * inv;
        const V3 qv{tv.y * e1.z - tv.z * e1.y, tv.z * e1.x - tv.x * e1.z, tv.x * e1.y - tv.y * e1.x};
        const float b2 = (dir.x * qv.x + dir.y * qv.y + dir.z * qv.z) * inv;
        const float t = (e2.x * qv.x + e2.y * qv.y + e2.z * qv.z) * inv;   // = camera z of the hit (dir.z == 1)
        const float b0 = 1.f - b1 - b2;
        u = b0 * uvs[2 * i0] + b1 * uvs[2 * i1] + b2 * uvs[2 * i2];
        v = b0 * uvs[2 * i0 + 1] + b1 * uvs[2 * i1 + 1] + b2 * uvs[2 * i2 + 1];
        dep = t;
        // interpolated vertex normal in camera space against the direction to the eye
        float nw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) nw[k] = b0 * normals[3 * (size_t)i0 + k] + b1 * normals[3 * (size_t)i1 + k] + b2 * normals[3 * (size_t)i2 + k];
        const V3 n{c.r[0] * nw[0] + c.r[1] * nw[1] + c.r[2] * nw[2], c.r[3] * nw[0] + c.r[4] * nw[1] + c.r[5] * nw[2],
                   c.r[6] * nw[0] + c.r[7] * nw[1] + c.r[8] * nw[2]};
        const float nl = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z), dl = sqrtf(dir.x * dir.x + dir.y * dir.y + 1.f);
        ang = (nl > 0.f) ? fmaxf(-(n.x * dir.x + n.y * dir.y + n.z) / (nl * dl), 0.f) : 0.f;
    }
    uv_out[3 * (size_t)i] = u;
    uv_out[3 * (size_t)i + 1] = v;
    uv_out[3 * (size_t)i + 2] = 0.f;   // the reference stores the mip level here; its loader drops the channel
    angle_out[i] = ang;
    depth_out[i] = dep;
}

// ---------------------------------------------------------------------------------------------------
// R2: textured re-render of a rasterised view - trilinear mip-mapped lookup of an RGB texture at the UV map
// (the reference's rgb shader path: GL_LINEAR_MIPMAP_LINEAR on a glGenerateMipmap pyramid, renderer.cpp:110-139,
// shader/rgb.frag; its fixed ambient / diffuse lighting factor and the anisotropic extension are not reproduced).
// Texel (i, j) of a W x H level is centred at ((i + 0.5) / W, (j + 0.5) / H); clamp to edge. The level of detail is
// log2 of the larger screen-space UV footprint (forward differences of the UV map, in texels of level 0).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mip_downsample_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hs,
                                                             int Ws, int Hd, int Wd) {
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (i >= Hd * Wd) return;
    const int y = i / Wd, x = i - y * Wd;
    const int y0 = min(2 * y, Hs - 1), y1 = min(2 * y + 1, Hs - 1), x0 = min(2 * x, Ws - 1), x1 = min(2 * x + 1, Ws - 1);
    const float* p = src + (size_t)c * Hs * Ws;
    dst[(size_t)c * Hd * Wd + i] = 0.25f * (p[y0 * Ws + x0] + p[y0 * Ws + x1] + p[y1 * Ws + x0] + p[y1 * Ws + x1]);
}

struct MipChain {
    const float* p[16];
    int w[16], h[16];
    int n;
};

__device__ __forceinline__ void mip_bilinear(const MipChain& m, int l, float u, float v, float out[3]) {
    const int W = m.w[l], H = m.h[l];
    const float x = fminf(fmaxf(u * W - 0.5f, 0.f), (float)(W - 1)), y = fminf(fmaxf(v * H - 0.5f, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(x), y0 = (int)floorf(y), x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float fx = x - x0, fy = y - y0;
    const size_t cs = (size_t)W * H;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = m.p[l] + c * cs;
        out[c] = (1.f - fy) * ((1.f - fx) * p[y0 * W + x0] + fx * p[y0 * W + x1]) +
                 fy * ((1.f - fx) * p[y1 * W + x0] + fx * p[y1 * W + x1]);
    }
}

__global__ __launch_bounds__(256) void tex_sample_mip_kernel(MipChain m, const float* __restrict__ uv, int H, int W,
                                                             float* __restrict__ out, float* __restrict__ lod_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const float u = uv[3 * (size_t)i], v = uv[3 * (size_t)i + 1];
    float rgb[3] = {0.f, 0.f, 0.f}, lod = 0.f;
    if (u != 0.f || v != 0.f) {   // background pixels carry uv = 0
        // footprint from the neighbours that also hit the surface (forward, else backward difference)
        auto duv = [&](int j, float& du, float& dv) {
            const float uu = uv[3 * (size_t)j], vv = uv[3 * (size_t)j + 1];
            const bool ok = uu != 0.f || vv != 0.f;
            du = ok ? uu - u : 0.f;
            dv = ok ? vv - v : 0.f;
            return ok;
        };
        float dux = 0.f, dvx = 0.f, duy = 0.f, dvy = 0.f;
        if (!(x + 1 < W && duv(i + 1, dux, dvx)) && x > 0) duv(i - 1, dux, dvx);
        if (!(y + 1 < H && duv(i + W, duy, dvy)) && y > 0) duv(i - W, duy, dvy);
        const float W0 = (float)m.w[0], H0 = (float)m.h[0];
        const float rx = sqrtf(dux * dux * W0 * W0 + dvx * dvx * H0 * H0), ry = sqrtf(duy * duy * W0 * W0 + dvy * dvy * H0 * H0);
        const float rho = fmaxf(rx, ry);
        lod = fminf(fmaxf(rho > 0.f ? log2f(rho) : 0.f, 0.f), (float)(m.n - 1));
        const int l0 = (int)floorf(lod), l1 = min(l0 + 1, m.n - 1);
        const float f = lod - l0;
        float a[3], b[3];
        mip_bilinear(m, l0, u, v, a);
        mip_bilinear(m, l1, u, v, b);
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = (1.f - f) * a[c] + f * b[c];
    }
    const size_t n = (size_t)H * W;
    out[i] = rgb[0];
    out[n + i] = rgb[1];
    out[2 * n + i] = rgb[2];
    if (lod_out) lod_out[i] = lod;
}

}  // namespace sm

extern "C" {

int sm_raster_maps(const float* verts, const float* normals, const float* uvs, const int32_t* faces, int n_faces,
                   const float* world2cam, const float* intrinsics, int H, int W, float znear, float zfar,
                   uint64_t* zbuf, float* big_scratch, int big_cap, float* uv_out, float* angle_out, float* depth_out,
                   void* stream) {
    if (n_faces < 0 || H < 1 || W < 1 || !(znear > 0.f) || !(zfar > znear)) return (int)hipErrorInvalidValue;
    sm::RasterCam c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.r[3 * i + j] = world2cam[4 * i + j];
        c.t[i] = world2cam[4 * i + 3];
    }
    c.fx = intrinsics[0]; c.fy = intrinsics[1]; c.cx = intrinsics[2]; c.cy = intrinsics[3];
    c.H = H; c.W = W; c.znear = znear; c.zfar = zfar;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(zbuf, 0xff, (size_t)H * W * sizeof(uint64_t), s);
    if (e != hipSuccess) return (int)e;
    if (n_faces > 0) {
        // big_scratch: [1 + 10 * big_cap] floats; word 0 is the queue counter
        sm::BigList big{nullptr, nullptr, 0};
        if (big_scratch && big_cap > 0) {
            big = sm::BigList{big_scratch + 1, reinterpret_cast<int*>(big_scratch), big_cap};
            e = hipMemsetAsync(big_scratch, 0, sizeof(float), s);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(sm::raster_depth_kernel, dim3((n_faces + 255) / 256), dim3(256), 0, s, c, verts, faces, n_faces,
                           reinterpret_cast<unsigned long long*>(zbuf), big);
        SM_LAUNCH_CHECK();
        if (big.tris) {
            hipLaunchKernelGGL(sm::raster_big_kernel, dim3(std::min(big_cap, 2048)), dim3(256), 0, s, c, big,
                               reinterpret_cast<unsigned long long*>(zbuf));
            SM_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(sm::raster_shade_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, c, verts, normals, uvs, faces,
                       reinterpret_cast<const unsigned long long*>(zbuf), uv_out, angle_out, depth_out);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_mip_downsample(const float* src, float* dst, int C, int Hs, int Ws, void* stream) {
    const int Hd = std::max(Hs / 2, 1), Wd = std::max(Ws / 2, 1);
    if (C < 1 || Hs < 1 || Ws < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sm::mip_downsample_kernel, dim3((Hd * Wd + 255) / 256, C), dim3(256), 0, (hipStream_t)stream, src, dst,
                       Hs, Ws, Hd, Wd);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_tex_sample_mip(const float* const* levels, const int* level_w, const int* level_h, int n_levels, const float* uv,
                      int H, int W, float* out, float* lod_out, void* stream) {
    if (n_levels < 1 || n_levels > 16) return (int)hipErrorInvalidValue;
    sm::MipChain m;
    m.n = n_levels;
    for (int l = 0; l < n_levels; ++l) { m.p[l] = levels[l]; m.w[l] = level_w[l]; m.h[l] = level_h[l]; }
    hipLaunchKernelGGL(sm::tex_sample_mip_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, m, uv, H, W, out,
                       lod_out);
    SM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
