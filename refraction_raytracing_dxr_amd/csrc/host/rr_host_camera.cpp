// rr_host_camera.cpp -- the per-frame camera constants of RefractionDemo::drawFrame
// (RefractionDemo.cpp:559-566), host side, no device.
//
// The reference computes them with DirectXMath (Windows SDK header, not in its tree):
//   proj  = XMMatrixPerspectiveFovLH(52/180*3.1415, 1.333, 1, 125)          :559
//   loc   = (5 cos a, 0, 5 sin a, 1)                                       :560
//   world = XMMatrixTranslationFromVector(loc)                             :561
//   view  = XMMatrixLookAtLH((cos -a, 0, sin -a), origin, +Y)              :562
//   proj_inv = XMMatrixInverse(proj * world * view)                        :563-565
// DirectXMath's published algorithms are restated here in fp32 with its row-vector convention.
#include "../../../include/rrdxr.h"

#include <cmath>
#include <cstring>

namespace {

struct Vec3 { float x, y, z; };
struct Mat4 {
    float r[4][4];
    static Mat4 zero() { Mat4 m; std::memset(&m, 0, sizeof m); return m; }
    static Mat4 identity() { Mat4 m = zero(); m.r[0][0] = m.r[1][1] = m.r[2][2] = m.r[3][3] = 1.0f; return m; }
};

Vec3 operator-(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
float dot(Vec3 a, Vec3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
Vec3 cross(Vec3 a, Vec3 b)
{
    return { fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)) };
}
Vec3 normalized(Vec3 a)
{
    float s = 1.0f / std::sqrt(dot(a, a));
    return { a.x * s, a.y * s, a.z * s };
}

// XMScalarSinCos: range reduction to [-pi/2, pi/2] + 11th/10th-degree minimax polynomials
void scalar_sin_cos(float v, float& s, float& c)
{
    float q = 0.159154943f * v;
    q = v >= 0.0f ? (float)(int)(q + 0.5f) : (float)(int)(q - 0.5f);
    float y = v - 6.283185307f * q;
    float sign = 1.0f;
    if (y > 1.570796327f) { y = 3.141592654f - y; sign = -1.0f; }
    else if (y < -1.570796327f) { y = -3.141592654f - y; sign = -1.0f; }
    const float y2 = y * y;
    s = (((((-2.3889859e-08f * y2 + 2.7525562e-06f) * y2 - 0.00019840874f) * y2 + 0.0083333310f) * y2 - 0.16666667f) * y2 + 1.0f) * y;
    c = sign * (((((-2.6051615e-07f * y2 + 2.4760495e-05f) * y2 - 0.0013888378f) * y2 + 0.041666638f) * y2 - 0.5f) * y2 + 1.0f);
}

// XMMatrixMultiply, SSE summation order: (x*m0 + z*m2) + (y*m1 + w*m3)
Mat4 mul(const Mat4& a, const Mat4& b)
{
    Mat4 o;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            o.r[i][j] = (a.r[i][0] * b.r[0][j] + a.r[i][2] * b.r[2][j]) + (a.r[i][1] * b.r[1][j] + a.r[i][3] * b.r[3][j]);
    return o;
}

// general 4x4 inverse by 2x2 sub-determinants (Laplace expansion over row pairs)
bool inverse(const Mat4& m, Mat4& out)
{
    const float (*a)[4] = m.r;
    const float s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1];
    const float s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2];
    const float s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3];
    const float s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2];
    const float s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3];
    const float s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
    const float c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3];
    const float c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
    const float c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
    const float c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
    const float c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2];
    const float c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    if (det == 0.0f || !std::isfinite(det)) return false;
    const float id = 1.0f / det;
    float (*b)[4] = out.r;
    b[0][0] = ( a[1][1] * c5 - a[1][2] * c4 + a[1][3] * c3) * id;
    b[0][1] = (-a[0][1] * c5 + a[0][2] * c4 - a[0][3] * c3) * id;
    b[0][2] = ( a[3][1] * s5 - a[3][2] * s4 + a[3][3] * s3) * id;
    b[0][3] = (-a[2][1] * s5 + a[2][2] * s4 - a[2][3] * s3) * id;
    b[1][0] = (-a[1][0] * c5 + a[1][2] * c2 - a[1][3] * c1) * id;
    b[1][1] = ( a[0][0] * c5 - a[0][2] * c2 + a[0][3] * c1) * id;
    b[1][2] = (-a[3][0] * s5 + a[3][2] * s2 - a[3][3] * s1) * id;
    b[1][3] = ( a[2][0] * s5 - a[2][2] * s2 + a[2][3] * s1) * id;
    b[2][0] = ( a[1][0] * c4 - a[1][1] * c2 + a[1][3] * c0) * id;
    b[2][1] = (-a[0][0] * c4 + a[0][1] * c2 - a[0][3] * c0) * id;
    b[2][2] = ( a[3][0] * s4 - a[3][1] * s2 + a[3][3] * s0) * id;
    b[2][3] = (-a[2][0] * s4 + a[2][1] * s2 - a[2][3] * s0) * id;
    b[3][0] = (-a[1][0] * c3 + a[1][1] * c1 - a[1][2] * c0) * id;
    b[3][1] = ( a[0][0] * c3 - a[0][1] * c1 + a[0][2] * c0) * id;
    b[3][2] = (-a[3][0] * s3 + a[3][1] * s1 - a[3][2] * s0) * id;
    b[3][3] = ( a[2][0] * s3 - a[2][1] * s1 + a[2][2] * s0) * id;
    return true;
}

} // namespace

extern "C" int rr_host_camera_orbit(float angle, float fov_y, float aspect, float zn, float zf, rr_scene_constants* out)
{
    if (!out || !(fov_y > 0.0f) || !(aspect > 0.0f) || zn == zf) return RR_ERR_INVALID_ARGUMENT;

    // XMMatrixPerspectiveFovLH
    float sf, cf;
    scalar_sin_cos(0.5f * fov_y, sf, cf);
    const float h = cf / sf, w = h / aspect, range = zf / (zf - zn);
    Mat4 proj = Mat4::zero();
    proj.r[0][0] = w; proj.r[1][1] = h; proj.r[2][2] = range; proj.r[2][3] = 1.0f; proj.r[3][2] = -range * zn;

    // camera_loc and the translation "world" matrix
    const float loc[4] = { 5 * cosf(angle), 0.0f, 5 * sinf(angle), 1.0f };
    Mat4 world = Mat4::identity();
    world.r[3][0] = loc[0]; world.r[3][1] = loc[1]; world.r[3][2] = loc[2];

    // XMMatrixLookAtLH -> XMMatrixLookToLH(eye, focus - eye, up)
    const Vec3 eye = { cosf(-angle), 0.0f, sinf(-angle) };
    const Vec3 up = { 0.0f, 1.0f, 0.0f };
    const Vec3 zaxis = normalized(Vec3{ 0.0f, 0.0f, 0.0f } - eye);
    const Vec3 xaxis = normalized(cross(up, zaxis));
    const Vec3 yaxis = cross(zaxis, xaxis);
    const Vec3 neg_eye = { -eye.x, -eye.y, -eye.z };
    Mat4 view = Mat4::identity();
    view.r[0][0] = xaxis.x; view.r[1][0] = xaxis.y; view.r[2][0] = xaxis.z; view.r[3][0] = dot(xaxis, neg_eye);
    view.r[0][1] = yaxis.x; view.r[1][1] = yaxis.y; view.r[2][1] = yaxis.z; view.r[3][1] = dot(yaxis, neg_eye);
    view.r[0][2] = zaxis.x; view.r[1][2] = zaxis.y; view.r[2][2] = zaxis.z; view.r[3][2] = dot(zaxis, neg_eye);

    Mat4 inv;
    if (!inverse(mul(mul(proj, world), view), inv)) return RR_ERR_INVALID_ARGUMENT;
    std::memcpy(out->proj_inv, inv.r, 64);
    std::memcpy(out->camera_loc, loc, 16);
    return RR_OK;
}
