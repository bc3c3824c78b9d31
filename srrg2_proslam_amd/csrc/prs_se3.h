// prs_se3.h -- exact float SE(3) / 6x6 solver helpers for the aligner kernels.
// Every expression is written as explicit two-operand operations in a fixed order and the library
// is compiled with -ffp-contract=off; division and sqrt are IEEE correctly rounded (hipcc default),
// so results are reproducible bit-for-bit against a plain sequential float evaluation.
// 4x4 transforms are row-major float[16].
#pragma once
#include <hip/hip_runtime.h>

namespace prs {

__device__ __forceinline__ void se3_identity(float* T) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  }
}

// Isometry inverse [R^T | -R^T t]
__device__ __forceinline__ void se3_inverse(const float* T, float* Ti) {
  const float tx = T[3], ty = T[7], tz = T[11];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float r0 = T[0 + i], r1 = T[4 + i], r2 = T[8 + i];  // row i of R^T
    Ti[4 * i + 0]  = r0;
    Ti[4 * i + 1]  = r1;
    Ti[4 * i + 2]  = r2;
    Ti[4 * i + 3]  = -((r0 * tx + r1 * ty) + r2 * tz);
  }
  Ti[12] = 0.0f;
  Ti[13] = 0.0f;
  Ti[14] = 0.0f;
  Ti[15] = 1.0f;
}

// C = A * B (C must not alias A or B)
__device__ __forceinline__ void se3_mul(const float* A, const float* B, float* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      C[4 * i + j] = (A[4 * i + 0] * B[0 + j] + A[4 * i + 1] * B[4 + j]) + A[4 * i + 2] * B[8 + j];
    }
    C[4 * i + 3] = ((A[4 * i + 0] * B[3] + A[4 * i + 1] * B[7]) + A[4 * i + 2] * B[11]) + A[4 * i + 3];
  }
  C[12] = 0.0f;
  C[13] = 0.0f;
  C[14] = 0.0f;
  C[15] = 1.0f;
}

// geometry3d::t2tnq (srrg2_core): translation + imaginary part of the normalised quaternion, w >= 0
// (used at CF/correspondence_finder_projective_base_impl.cpp:182)
__device__ __forceinline__ void t2tnq(const float* T, float* v6) {
  const float m00 = T[0], m01 = T[1], m02 = T[2];
  const float m10 = T[4], m11 = T[5], m12 = T[6];
  const float m20 = T[8], m21 = T[9], m22 = T[10];
  float q0, q1, q2, q3;  // w x y z
  float t = (m00 + m11) + m22;
  if (t > 0.0f) {
    t  = sqrtf(t + 1.0f);
    q0 = 0.5f * t;
    t  = 0.5f / t;
    q1 = (m21 - m12) * t;
    q2 = (m02 - m20) * t;
    q3 = (m10 - m01) * t;
  } else {
    int i = 0;
    if (m11 > m00) {
      i = 1;
    }
    if (m22 > (i == 0 ? m00 : m11)) {
      i = 2;
    }
    if (i == 0) {  // j = 1, k = 2
      t  = sqrtf(((m00 - m11) - m22) + 1.0f);
      q1 = 0.5f * t;
      t  = 0.5f / t;
      q0 = (m21 - m12) * t;
      q2 = (m10 + m01) * t;
      q3 = (m20 + m02) * t;
    } else if (i == 1) {  // j = 2, k = 0
      t  = sqrtf(((m11 - m22) - m00) + 1.0f);
      q2 = 0.5f * t;
      t  = 0.5f / t;
      q0 = (m02 - m20) * t;
      q3 = (m21 + m12) * t;
      q1 = (m01 + m10) * t;
    } else {  // j = 0, k = 1
      t  = sqrtf(((m22 - m00) - m11) + 1.0f);
      q3 = 0.5f * t;
      t  = 0.5f / t;
      q0 = (m10 - m01) * t;
      q1 = (m02 + m20) * t;
      q2 = (m12 + m21) * t;
    }
  }
  const float n = sqrtf(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3);
  float s       = 1.0f / n;
  if (q0 < 0.0f) {
    s = -s;
  }
  v6[0] = T[3];
  v6[1] = T[7];
  v6[2] = T[11];
  v6[3] = q1 * s;
  v6[4] = q2 * s;
  v6[5] = q3 * s;
}

// perturbation [dt; dq] -> isometry, q = (sqrt(1 - |dq|^2), dq) (VariableSE3QuaternionRight)
__device__ __forceinline__ void tnq2t(const float* v6, float* T) {
  float x = v6[3], y = v6[4], z = v6[5];
  const float n2 = (x * x + y * y) + z * z;
  float w;
  if (n2 < 1.0f) {
    w = sqrtf(1.0f - n2);
  } else {
    const float s = 1.0f / sqrtf(n2);
    x *= s;
    y *= s;
    z *= s;
    w = 0.0f;
  }
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  T[0]  = 1.0f - (tyy + tzz);
  T[1]  = txy - twz;
  T[2]  = txz + twy;
  T[3]  = v6[0];
  T[4]  = txy + twz;
  T[5]  = 1.0f - (txx + tzz);
  T[6]  = tyz - twx;
  T[7]  = v6[1];
  T[8]  = txz - twy;
  T[9]  = tyz + twx;
  T[10] = 1.0f - (txx + tyy);
  T[11] = v6[2];
  T[12] = 0.0f;
  T[13] = 0.0f;
  T[14] = 0.0f;
  T[15] = 1.0f;
}

// The aligner sums its normal equations in the CAMERA frame (align.hip, factor_accumulate): with A = [R | t] the transform points
// go through, J = D G_c Rt, G_c = [ wt I | -[y]x ], y = 2 R p, Rt = blockdiag(R, R), so J^T Omega J = Rt^T (G_c^T D^T Omega D G_c) Rt
// and the rotation is the same for every correspondence: it is applied once, here, to the summed system.  Row r of the result
// (r = 3 * blk + i): v = (column i of R)^T Y, out = v R for the two 3 x 3 blocks Y of block row blk of H; b likewise.  The rotated
// matrix is symmetric up to rounding; its LOWER triangle (row >= column) is the system and is mirrored into the upper one.
// In place; gn_kernel evaluates the same expressions with one lane per row.
__device__ __forceinline__ void rotate_normal_equations(const float* A, float* H, float* b) {
  float Hn[36], bn[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int blk = r >= 3 ? 1 : 0;
    const int i   = r - 3 * blk;
    const float Ri0 = A[i], Ri1 = A[4 + i], Ri2 = A[8 + i];
    const float* Y = H + 18 * blk;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c] = fmaf(Ri2, Y[12 + 3 * cb + c], fmaf(Ri1, Y[6 + 3 * cb + c], Ri0 * Y[3 * cb + c]));
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Hn[6 * r + 3 * cb + j] = fmaf(v[2], A[8 + j], fmaf(v[1], A[4 + j], v[0] * A[j]));
      }
    }
    bn[r] = fmaf(Ri2, b[3 * blk + 2], fmaf(Ri1, b[3 * blk + 1], Ri0 * b[3 * blk]));
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      H[6 * r + c] = Hn[6 * r + c];
      H[6 * c + r] = Hn[6 * r + c];
    }
    b[r] = bn[r];
  }
}

// (H + damping diag(H)) dx = -b by LDL^T (L unit lower triangular, D diagonal; the damping form is a round-4 result of
// tools/sweep_a13.py; identity_damping selects (H + damping I) dx = -b, prs_aligner_params.damping_form).  H: full 6x6 row-major, the LOWER triangle is read.  U[i][j] = L[i][j] * D[j] is the entry before its division:
//   d_j = (1 + damping) H_jj - sum_k<j L_jk U_jk;   U_ij = H_ij - sum_k<j L_ik U_jk,  L_ij = U_ij / d_j   (one reciprocal per pivot)
//   y = L^-1 (-b);   dx = L^-T (y / d)
// Fused multiply-subtracts in the order written; returns false when a pivot is not positive (dx is then garbage, to be dropped).
// No square roots: round 4 replaced the Cholesky factorisation (six correctly rounded sqrt + six divisions were a third of the
// instructions of the solve).  gn_kernel evaluates this function on every lane of the solving wave (uniform), the CPU checker
// restates it (orc_gn_step).
__device__ __forceinline__ bool ldlt_solve6(const float* H, const float* b, const float damping, float* dx, const bool identity_damping = false) {
  float L[6][6], U[6][6], inv[6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = identity_damping ? H[6 * j + j] + damping : fmaf(damping, H[6 * j + j], H[6 * j + j]);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k < j) {
        d = fmaf(-L[j][k], U[j][k], d);
      }
    }
    ok     = ok && d > 0.0f;  // (no early exit: a failed pivot only poisons values that are dropped)
    inv[j] = recip_exact(d);  // (= 1.0f / d, bit for bit: prs_device.h)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i > j) {
        float v = H[6 * i + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          if (k < j) {
            v = fmaf(-L[i][k], U[j][k], v);
          }
        }
        U[i][j] = v;
        L[i][j] = v * inv[j];
      }
    }
  }
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float v = -b[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k < i) {
        v = fmaf(-L[i][k], y[k], v);
      }
    }
    y[i] = v;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float v = y[i] * inv[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k > i) {
        v = fmaf(-L[k][i], dx[k], v);
      }
    }
    dx[i] = v;
  }
  return ok;
}

// one damped Gauss-Newton step: X <- X * exp(dx); returns false (X untouched) when the system is not positive definite
__device__ __forceinline__ bool gn_step(const float* H, const float* b, float damping, float* X, const bool identity_damping = false, float* dx_out = nullptr) {
  float dx[6];
  const bool ok = ldlt_solve6(H, b, damping, dx, identity_damping);
  if (dx_out) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dx_out[i] = dx[i];
    }
  }
  float D[16], Xn[16];
  tnq2t(dx, D);
  se3_mul(X, D, Xn);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    X[i] = ok ? Xn[i] : X[i];
  }
  return ok;
}

}  // namespace prs
