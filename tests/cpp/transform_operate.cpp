// lrl::Transform::operate (raytracing_amd/host/lrl/RayTracing.hpp): the interactive camera of the demos.  The reference's key / mouse
// maths is not recoverable (SURVEY App. A "Not recovered"), so this is the library's own; what is pinned here are its properties:
// keys translate in the camera frame, the scroll wheel dollies along the view axis and dies away, a left drag rotates by |drag| / depth
// about an axis in the image plane, the rotation stays orthonormal, the Trans record follows, and the dirty flags drive the frame loop
// (transUniform.refreshData / decayOriginCalc.run, BVH/BVH.cpp:733-738).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>

#include "lrl/RayTracing.hpp"

using namespace lrl;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); ++failures; } } while (0)
static bool near(double a, double b, double eps = 1e-12) { return std::fabs(a - b) <= eps; }

static bool orthonormal(Math::mat3<double> const& m) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double d = 0;
            for (int k = 0; k < 3; ++k) d += m.data[k][i] * m.data[k][j];
            if (!near(d, i == j ? 1.0 : 0.0, 1e-12)) return false;
        }
    const double det = m.data[0][0] * (m.data[1][1] * m.data[2][2] - m.data[1][2] * m.data[2][1]) -
                       m.data[0][1] * (m.data[1][0] * m.data[2][2] - m.data[1][2] * m.data[2][0]) +
                       m.data[0][2] * (m.data[1][0] * m.data[2][1] - m.data[1][1] * m.data[2][0]);
    return near(det, 1.0, 1e-12);
}

int main() {
    Transform t({{60.0}, {0.09, 0.9, 0.01}, {0.15}, {0, 0, 30}, 700.0});      // the reference's constructor arguments (BVH/BVH.cpp:245)
    FrameScale size; size.w = 720; size.h = 480;
    t.init(size);
    CHECK(near(t.bufferData.data.z0, -float(480.0 / std::tan(60.0 * Math::Pi / 360.0)), 1e-3));      // z0 = -H / tan(fovy / 2)
    CHECK(t.bufferData.data.r0[2] == 30.0f && t.bufferData.data.trans[0][0] == 1.0f);
    t.moved = false; t.updated = false;

    t.operate();                                   // no input: nothing changes, nothing is flagged
    CHECK(!t.moved && !t.updated && t.dr[0] == 0 && t.dr[2] == 30);

    t.key.refresh(1, true);                        // "right" held: +x of the camera frame, key ratio per frame
    t.operate(); t.operate();
    CHECK(near(t.dr[0], 0.30) && near(t.dr[1], 0) && near(t.dr[2], 30));
    CHECK(t.moved && t.updated && near(t.bufferData.data.r0[0], 0.30f, 1e-6));
    t.key.refresh(1, false);
    t.key.refresh(2, true);                        // "up": +y
    t.operate();
    CHECK(near(t.dr[1], 0.15));
    t.key.refresh(2, false);

    t.scroll.refresh(1.0);                         // one notch: dolly forward (-z), then the motion dies away by decreaseRatio
    const double z0 = t.dr[2];
    t.operate();
    CHECK(near(t.dr[2], z0 - 0.09));
    t.operate();
    CHECK(near(t.dr[2], z0 - 0.09 - 0.081));
    for (int i = 0; i < 100; ++i) t.operate();
    CHECK(t.scroll.total == 0);                    // below the threshold the wheel stops
    const double zEnd = t.dr[2];
    t.moved = false;
    t.operate();
    CHECK(t.dr[2] == zEnd && !t.moved);

    t.mouse.refreshButton(0, true);                // left drag of 70 pixels along x: a rotation by 70 / depth about the image's y axis
    t.mouse.refreshPos(100, 100);
    t.mouse.refreshPos(170, 100);
    t.operate();
    CHECK(orthonormal(t.trans));
    const double ang = 70.0 / 700.0;
    CHECK(near(t.trans.data[0][0], std::cos(ang)) && near(t.trans.data[1][1], 1.0) && near(std::fabs(t.trans.data[0][2]), std::sin(ang)));
    CHECK(near(t.bufferData.data.trans[0][0], float(std::cos(ang)), 1e-7) && t.moved);
    for (int i = 0; i < 500; ++i) { t.mouse.refreshPos(170 + 3 * i, 100 + 2 * i); t.operate(); }
    CHECK(orthonormal(t.trans));                   // no drift away from a rotation after hundreds of increments
    t.mouse.refreshButton(0, false);
    t.mouse.refreshPos(0, 0);
    t.moved = false;
    t.operate();
    CHECK(!t.moved);                               // moving the pointer without the button does nothing

    // keys act in the CAMERA frame: after the rotation "right" no longer moves along world x alone
    const Math::vec3<double> before = t.dr;
    t.key.refresh(1, true);
    t.operate();
    const double step[3] = {t.dr[0] - before[0], t.dr[1] - before[1], t.dr[2] - before[2]};
    CHECK(near(std::sqrt(step[0] * step[0] + step[1] * step[1] + step[2] * step[2]), 0.15, 1e-12));
    CHECK(near(step[0], 0.15 * t.trans.data[0][0], 1e-12) && near(step[2], 0.15 * t.trans.data[2][0], 1e-12));

    // the sub-pixel jitter as a shear of trans: S (x, y, z0) = (x + 2 dx, y + 2 dy, z0)
    LrlTrans id{};
    for (int i = 0; i < 3; ++i) id.trans[i][i] = 1.0f;
    id.z0 = -100.0f;
    const LrlTrans sh = Transform::sheared(id, 0.25f, -0.5f);
    const float v[3] = {3.0f, 4.0f, id.z0};
    float o[3];
    for (int i = 0; i < 3; ++i) o[i] = sh.trans[i][0] * v[0] + sh.trans[i][1] * v[1] + sh.trans[i][2] * v[2];
    CHECK(near(o[0], 3.5, 1e-5) && near(o[1], 3.0, 1e-5) && near(o[2], id.z0, 1e-5));

    if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
    std::printf("transform_operate: ok\n");
    return 0;
}
