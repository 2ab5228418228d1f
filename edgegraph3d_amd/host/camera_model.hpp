// Camera construction shared by the synthetic generator and the OpenMVG reader.
// Mirrors how the reference builds CameraType from an OpenMVG file
// (external/manifoldReconstructor/src/OpenMvgParser.cpp:107-125 cameraMatrix = eMatrix*kMatrix,
//  :289 translation = -center * rotation), including glm's float evaluation order
// (external/glm/glm/detail/type_mat4x4.inl:686-700, type_mat3x3.inl vec*mat), so the
// float cameraMatrix values are the ones the reference would hold.
#pragma once
#include <cstring>

namespace eg3dh {

// t_i = R(i,0)*(-C0) + R(i,1)*(-C1) + R(i,2)*(-C2)   (glm vec3 * mat3, left to right)
static inline void translation_from_center(const float R[9], const float C[3], float t[3]) {
  float v0 = -C[0], v1 = -C[1], v2 = -C[2];
  for (int i = 0; i < 3; i++) t[i] = (R[3 * i + 0] * v0 + R[3 * i + 1] * v1) + R[3 * i + 2] * v2;
}

// P = K4 * E4 with K4 = [K 0; 0 0] and E4 = [R t; 0 0 0 1], float, summed over k ascending.
// Row 3 is all zero (Q6).
static inline void camera_matrix(float focal, float ppx, float ppy, const float R[9], const float t[3], float P[16]) {
  float K[4][4], E[4][4];
  memset(K, 0, sizeof(K));
  memset(E, 0, sizeof(E));
  K[0][0] = focal;
  K[1][1] = focal;
  K[0][2] = ppx;
  K[1][2] = ppy;
  K[2][2] = 1.0f;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) E[r][c] = R[3 * r + c];
    E[r][3] = t[r];
  }
  E[3][3] = 1.0f;
  for (int c = 0; c < 4; c++)
    for (int j = 0; j < 4; j++) {
      float s = K[c][0] * E[0][j];
      s = s + K[c][1] * E[1][j];
      s = s + K[c][2] * E[2][j];
      s = s + K[c][3] * E[3][j];
      P[4 * c + j] = s;
    }
}

// Analytic fundamental matrix, double: l_j = F_ij * x_i with
// F_ij = K_j^-T [t_ij]x R_ij K_i^-1, R_ij = R_j R_i^T, t_ij = t_j - R_ij t_i.
static inline void fundamental_from_cameras(float fi, float pxi, float pyi, const float Ri[9], const float ti[3],
                                            float fj, float pxj, float pyj, const float Rj[9], const float tj[3],
                                            double F[9]) {
  double Rij[3][3], tij[3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += (double)Rj[3 * r + k] * (double)Ri[3 * c + k];
      Rij[r][c] = s;
    }
  for (int r = 0; r < 3; r++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += Rij[r][k] * (double)ti[k];
    tij[r] = (double)tj[r] - s;
  }
  double Tx[3][3] = {{0, -tij[2], tij[1]}, {tij[2], 0, -tij[0]}, {-tij[1], tij[0], 0}};
  double E[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Tx[r][k] * Rij[k][c];
      E[r][c] = s;
    }
  // K^-1 = [[1/f,0,-px/f],[0,1/f,-py/f],[0,0,1]]
  double Kii[3][3] = {{1.0 / fi, 0, -(double)pxi / fi}, {0, 1.0 / fi, -(double)pyi / fi}, {0, 0, 1}};
  double Kji[3][3] = {{1.0 / fj, 0, -(double)pxj / fj}, {0, 1.0 / fj, -(double)pyj / fj}, {0, 0, 1}};
  double T[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += E[r][k] * Kii[k][c];
      T[r][c] = s;
    }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Kji[k][r] * T[k][c];  // K_j^-T
      F[3 * r + c] = s;
    }
}

}  // namespace eg3dh
