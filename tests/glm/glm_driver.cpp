// glm_driver.cpp — TEST-ONLY, BUILD CONTAINER ONLY. Evaluates, with the header-only glm the reference vendors
// (/root/reference/external/glm, 0.9.6; given with -I on the command line, nothing of it is copied), the handful of
// glm expressions the hot path's arithmetic rests on, so that the oracle's hand-written evaluation orders (and the
// product's camera model) can be pinned bit for bit against the reference's own math library:
//   proj     vec4(X,1) * mat4, then x/z, y/z                 geometric_utilities.cpp:973-977 (compute_projection)
//   cam      t = -center * R (vec3 * mat3); P = E4 * K4      OpenMvgParser.cpp:289, :107-125 (cameraMatrix = eMatrix*kMatrix)
//   anglecos dot(a,b) / sqrt(dot(a,a) * dot(b,b)) on vec2    geometric_utilities.cpp:579-581, 590-618 (compute_anglecos)
//   mindist  clamp(dot(p-v, w-v) / l2), v + t * (w - v)      geometric_utilities.cpp:940-954 (minimum_distancesq);
//            squared_2d_distance = pow(float,2) sums         :555-557
//   ratio    a + ratio * (b - a) on vec2                     geometric_utilities.cpp:1370-1372
// I/O: stdin = mode name on argv[1], then raw little-endian float32 records on stdin; stdout = raw float32 results.
// Compiled with the reference's release flags (CMakeLists.txt:48: -O3 -funroll-loops, baseline x86-64: no FMA).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <glm/glm.hpp>

static bool rd(float* p, size_t n) { return fread(p, sizeof(float), n, stdin) == n; }
static void wr(const float* p, size_t n) { fwrite(p, sizeof(float), n, stdout); }

// the reference squares through pow(float, int) -> double and rounds the SUM once (geometric_utilities.cpp:555-557)
static float sq2d(const glm::vec2& a, const glm::vec2& b) { return pow(a[0] - b[0], 2) + pow(a[1] - b[1], 2); }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const char* mode = argv[1];
  float in[32], out[32];
  if (!strcmp(mode, "proj")) {  // in: M[4][4] as filled [row][col] (16), X (3); out: u, v, w0, w1, w2
    while (rd(in, 19)) {
      glm::mat4 M;
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) M[r][c] = in[4 * r + c];
      const glm::vec3 X(in[16], in[17], in[18]);
      const glm::vec4 h = glm::vec4(X[0], X[1], X[2], 1.0) * M;
      const glm::vec2 uv(h[0] / h[2], h[1] / h[2]);
      out[0] = uv[0];
      out[1] = uv[1];
      out[2] = h[0];
      out[3] = h[1];
      out[4] = h[2];
      wr(out, 5);
    }
  } else if (!strcmp(mode, "cam")) {  // in: focal, ppx, ppy, R[3][3] (9), C (3); out: t (3), P[4][4] (16)
    while (rd(in, 15)) {
      glm::mat3 K(0.0f), R;
      K[0][0] = in[0];
      K[1][1] = in[0];
      K[0][2] = in[1];
      K[1][2] = in[2];
      K[2][2] = 1.0f;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[r][c] = in[3 + 3 * r + c];
      const glm::vec3 C(in[12], in[13], in[14]);
      const glm::vec3 t = -C * R;
      glm::mat4 E(0.0), K4(0.0);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
          E[r][c] = R[r][c];
          K4[r][c] = K[r][c];
        }
      E[0][3] = t[0];
      E[1][3] = t[1];
      E[2][3] = t[2];
      E[3][3] = 1.0;
      const glm::mat4 P = E * K4;
      out[0] = t[0];
      out[1] = t[1];
      out[2] = t[2];
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[3 + 4 * r + c] = P[r][c];
      wr(out, 19);
    }
  } else if (!strcmp(mode, "anglecos")) {  // in: segment x1 y1 x2 y2, line a b c; out: cos
    while (rd(in, 7)) {
      const glm::vec2 a(in[2] - in[0], in[3] - in[1]);
      glm::vec2 b;
      if (in[5] == 0)
        b = glm::vec2(0.0, 1.0);
      else
        b = glm::vec2(1.0, -in[4] / in[5]);
      out[0] = glm::dot(a, b) / sqrt(glm::dot(a, a) * glm::dot(b, b));
      wr(out, 1);
    }
  } else if (!strcmp(mode, "mindist")) {  // in: p, v, w; out: d2, proj x, proj y
    while (rd(in, 6)) {
      const glm::vec2 p(in[0], in[1]), v(in[2], in[3]), w(in[4], in[5]);
      glm::vec2 projection;
      float d2;
      const float l2 = sq2d(v, w);
      if (l2 == 0.0) {
        projection = v;
        d2 = sq2d(p, v);
      } else {
        const float t = std::max<float>(0, std::min<float>(1, glm::dot(p - v, w - v) / l2));
        projection = v + t * (w - v);
        d2 = sq2d(p, projection);
      }
      out[0] = d2;
      out[1] = projection[0];
      out[2] = projection[1];
      wr(out, 3);
    }
  } else if (!strcmp(mode, "ratio")) {  // in: a, b, ratio; out: x, y
    while (rd(in, 5)) {
      const glm::vec2 a(in[0], in[1]), b(in[2], in[3]);
      const glm::vec2 r = a + in[4] * (b - a);
      out[0] = r[0];
      out[1] = r[1];
      wr(out, 2);
    }
  } else {
    return 2;
  }
  return 0;
}
