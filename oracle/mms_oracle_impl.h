/*
 * oracle/mms_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see mms_oracle.c header).
 *
 * Type-generic body of the CPU restatement.  Included twice by mms_oracle.c,
 * once with Dtype=float / SFX(x)=x##_f32 and once with Dtype=double / _f64,
 * mirroring the reference's INSTANTIATE_CLASS(float, double)
 * (reference include/caffe/common.hpp:41-44).
 *
 * Every function cites the reference file:line whose loop nest, evaluation
 * order and type promotions it restates.  Paths are relative to the reference
 * checkout (src/caffe/layers/...).
 */

/* ---------------------------------------------------------------------------
 * BLAS stand-ins.  The reference calls CBLAS through
 * src/caffe/util/math_functions.cpp:13-58 (gemm), :341-355 (dot).  The provider
 * (MKL, Makefile.config:33) is not vendored, so the summation order inside
 * sdot/sgemm is not defined by the reference.  We use the plain k-ascending
 * order -- one legal BLAS ordering; tests hold BLAS-backed results to 1e-5,
 * not bit-exactness.
 * ------------------------------------------------------------------------- */
static Dtype SFX(o_dot)(int n, const Dtype* x, const Dtype* y) {
  Dtype s = 0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

/* Row-major C(MxN) = alpha*op(A)*op(B) + beta*C, op = transpose iff tX != 0.
 * beta == 0 overwrites C without reading it (BLAS semantics). */
static void SFX(o_gemm)(int tA, int tB, int M, int N, int K, Dtype alpha,
                        const Dtype* A, const Dtype* B, Dtype beta, Dtype* C) {
  for (int i = 0; i < M; ++i) {
    for (int j = 0; j < N; ++j) {
      Dtype s = 0;
      for (int k = 0; k < K; ++k) {
        Dtype av = tA ? A[(size_t)k * M + i] : A[(size_t)i * K + k];
        Dtype bv = tB ? B[(size_t)j * K + k] : B[(size_t)k * N + j];
        s += av * bv;
      }
      Dtype* c = &C[(size_t)i * N + j];
      if (beta == (Dtype)0) *c = alpha * s;
      else *c = alpha * s + beta * (*c);
    }
  }
}

/* ---------------------------------------------------------------------------
 * SimCross forward.  Reference: sim_cross_layer.cpp:83-163.
 *   q (N,W1,D), a (N,W2,D) row-major; top (N, M|1, W1, W2).
 *   mode 1 (:96-111)  top = 1/(1+sqrt(sum_d (q-a)^2)), d ascending, Dtype acc.
 *   mode 0 (:112-139) norms = sqrt(dot(x,x)) cached; top = dot(q,a)/n0/n1
 *                     (two successive divisions, :135).
 *   mode 2 (:140-161) per (n,m): tmp = Q_n W_m ; top = tmp A_n^T ; + bias.
 * norm0 (N,W1), norm1 (N,W2) are the layer's data{0,1}_norm_ scratch blobs
 * (mode 0 only; may be NULL otherwise).  W (M,D,D), bias (M,W1,W2) or NULL.
 * ------------------------------------------------------------------------- */
void SFX(oracle_simcross_forward)(int mode, int N, int W1, int W2, int D, int M,
                                  const Dtype* q, const Dtype* a,
                                  const Dtype* W, const Dtype* bias,
                                  Dtype* top, Dtype* norm0, Dtype* norm1) {
  if (mode == 1) {
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < W1; ++j)
        for (int k = 0; k < W2; ++k) {
          Dtype dist = 0;
          for (int dd = 0; dd < D; ++dd) {
            Dtype diff = q[((size_t)i * W1 + j) * D + dd] -
                         a[((size_t)i * W2 + k) * D + dd];
            dist += diff * diff;
          }
          dist = SQRT(dist);
          top[((size_t)i * W1 + j) * W2 + k] = 1 / (1 + dist);
        }
  } else if (mode == 0) {
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < W1; ++j) {
        const Dtype* x = q + ((size_t)i * W1 + j) * D;
        norm0[(size_t)i * W1 + j] = SQRT(SFX(o_dot)(D, x, x));
      }
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < W2; ++j) {
        const Dtype* x = a + ((size_t)i * W2 + j) * D;
        norm1[(size_t)i * W2 + j] = SQRT(SFX(o_dot)(D, x, x));
      }
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < W1; ++j)
        for (int k = 0; k < W2; ++k)
          top[((size_t)i * W1 + j) * W2 + k] =
              SFX(o_dot)(D, q + ((size_t)i * W1 + j) * D,
                         a + ((size_t)i * W2 + k) * D) /
              norm0[(size_t)i * W1 + j] / norm1[(size_t)i * W2 + k];
  } else if (mode == 2) {
    Dtype* tmp = (Dtype*)malloc(sizeof(Dtype) * (size_t)W1 * D);
    for (int i = 0; i < N; ++i) {
      for (int j = 0; j < M; ++j) {
        SFX(o_gemm)(0, 0, W1, D, D, (Dtype)1, q + (size_t)i * W1 * D,
                    W + (size_t)j * D * D, (Dtype)0, tmp);
        SFX(o_gemm)(0, 1, W1, W2, D, (Dtype)1, tmp, a + (size_t)i * W2 * D,
                    (Dtype)0, top + ((size_t)i * M + j) * W1 * W2);
      }
      if (bias) {
        Dtype* t = top + (size_t)i * M * W1 * W2;
        for (int e = 0; e < M * W1 * W2; ++e) t[e] = bias[e] + t[e];
      }
    }
    free(tmp);
  }
}

/* ---------------------------------------------------------------------------
 * SimCross backward.  Reference: sim_cross_layer.cpp:166-307.
 *   :176-177  both bottom diffs are zeroed unconditionally;
 *   :201      if either propagate_down flag is set BOTH are computed;
 *   mode 1 (:208-225): dd outermost; tt (Dtype) =
 *        dT*T*T*T*(q-a) / (T - 1 + 1e-9)   -- the 1e-9 literal is a double,
 *        so the divisor and the division are double for Dtype=float too;
 *        dq += tt (k index ascending), da += -tt (j index ascending).
 *   mode 0 (:226-250): norms from forward.
 *   mode 2 (:251-305): W.diff zeroed here (:256); six gemms per (n,m);
 *        bias.diff accumulated, not zeroed (:301-304).
 * dW (M,D,D), dbias (M,W1,W2): the parameter diff buffers (dbias carries its
 * previous contents in, like Blob::cpu_diff()).
 * ------------------------------------------------------------------------- */
void SFX(oracle_simcross_backward)(int mode, int N, int W1, int W2, int D, int M,
                                   const Dtype* q, const Dtype* a,
                                   const Dtype* W, int bias_term,
                                   const Dtype* top, const Dtype* top_diff,
                                   const Dtype* norm0, const Dtype* norm1,
                                   int propagate_down0, int propagate_down1,
                                   Dtype* dq, Dtype* da, Dtype* dW,
                                   Dtype* dbias) {
  memset(dq, 0, sizeof(Dtype) * (size_t)N * W1 * D);
  memset(da, 0, sizeof(Dtype) * (size_t)N * W2 * D);
  if (!(propagate_down0 || propagate_down1)) return;

  if (mode == 1) {
    for (int dd = 0; dd < D; ++dd)
      for (int j = 0; j < N; ++j)
        for (int k = 0; k < W1; ++k)
          for (int m = 0; m < W2; ++m) {
            size_t t = ((size_t)j * W1 + k) * W2 + m;
            size_t b0 = ((size_t)j * W1 + k) * D + dd;
            size_t b1 = ((size_t)j * W2 + m) * D + dd;
            Dtype tt = top_diff[t] * top[t] * top[t] * top[t] *
                       (q[b0] - a[b1]) / (top[t] - 1 + 1e-9);
            dq[b0] += tt;
            da[b1] += -tt;
          }
  } else if (mode == 0) {
    for (int dd = 0; dd < D; ++dd)
      for (int j = 0; j < N; ++j)
        for (int k = 0; k < W1; ++k)
          for (int m = 0; m < W2; ++m) {
            size_t t = ((size_t)j * W1 + k) * W2 + m;
            size_t b0 = ((size_t)j * W1 + k) * D + dd;
            size_t b1 = ((size_t)j * W2 + m) * D + dd;
            const Dtype nrm0 = norm0[(size_t)j * W1 + k];
            const Dtype nrm1 = norm1[(size_t)j * W2 + m];
            Dtype tt = top_diff[t] * (a[b1] / nrm0 / nrm1 -
                                      q[b0] * top[t] / (nrm0 * nrm0));
            dq[b0] += tt;
            tt = top_diff[t] * (q[b0] / nrm0 / nrm1 -
                                a[b1] * top[t] / (nrm1 * nrm1));
            da[b1] += tt;
          }
  } else if (mode == 2) {
    memset(dW, 0, sizeof(Dtype) * (size_t)M * D * D);
    size_t wmax = (size_t)(W1 > W2 ? W1 : W2);
    Dtype* temp0 = (Dtype*)malloc(sizeof(Dtype) * (size_t)W1 * D);
    Dtype* temp1 = (Dtype*)malloc(sizeof(Dtype) * wmax * D);
    for (int i = 0; i < N; ++i) {
      for (int j = 0; j < M; ++j) {
        const Dtype* dT = top_diff + ((size_t)i * M + j) * W1 * W2;
        const Dtype* Q = q + (size_t)i * W1 * D;
        const Dtype* A = a + (size_t)i * W2 * D;
        const Dtype* Wj = W + (size_t)j * D * D;
        /* :286-289  dW_j += (Q^T dT) A */
        SFX(o_gemm)(1, 0, D, W2, W1, (Dtype)1, Q, dT, (Dtype)0, temp1);
        SFX(o_gemm)(0, 0, D, D, W2, (Dtype)1, temp1, A, (Dtype)1,
                    dW + (size_t)j * D * D);
        /* :291-294  dQ += dT (W_j A^T)^T */
        SFX(o_gemm)(0, 1, D, W2, D, (Dtype)1, Wj, A, (Dtype)0, temp1);
        SFX(o_gemm)(0, 1, W1, D, W2, (Dtype)1, dT, temp1, (Dtype)1,
                    dq + (size_t)i * W1 * D);
        /* :296-299  dA += dT^T (Q W_j) */
        SFX(o_gemm)(0, 0, W1, D, D, (Dtype)1, Q, Wj, (Dtype)0, temp0);
        SFX(o_gemm)(1, 0, W2, D, W1, (Dtype)1, dT, temp0, (Dtype)1,
                    da + (size_t)i * W2 * D);
      }
      if (bias_term) {
        const Dtype* dT = top_diff + (size_t)i * M * W1 * W2;
        for (int e = 0; e < M * W1 * W2; ++e) dbias[e] = dT[e] + dbias[e];
      }
    }
    free(temp0);
    free(temp1);
  }
}

/* ---------------------------------------------------------------------------
 * SimMatrix.  Reference: sim_matrix_layer.cpp:53-65 (forward), :68-95 (backward).
 *   forward: tmp = Q W  (N x K2 x K1 gemm) written into bottom[1]'s DIFF buffer
 *            (:58 -- the scribble is observable, so `a_diff_scratch` is an
 *            output here too); top[i] = dot(a_i, tmp_i).
 *   backward: dW += dT_i * q_i a_i^T (N sger's, accumulating, only if
 *            param_propagate_down); dq_i = dT_i * W a_i ; da_i = dT_i * W^T q_i
 *            (gemv with beta = 0: overwrite), each only if propagate_down[i].
 * ------------------------------------------------------------------------- */
void SFX(oracle_simmatrix_forward)(int N, int K1, int K2, const Dtype* q,
                                   const Dtype* a, const Dtype* W, Dtype* top,
                                   Dtype* a_diff_scratch) {
  SFX(o_gemm)(0, 0, N, K2, K1, (Dtype)1, q, W, (Dtype)0, a_diff_scratch);
  for (int i = 0; i < N; ++i)
    top[i] = SFX(o_dot)(K2, a + (size_t)i * K2, a_diff_scratch + (size_t)i * K2);
}

void SFX(oracle_simmatrix_backward)(int N, int K1, int K2, const Dtype* q,
                                    const Dtype* a, const Dtype* W,
                                    const Dtype* top_diff,
                                    int param_propagate_down,
                                    int propagate_down0, int propagate_down1,
                                    Dtype* dq, Dtype* da, Dtype* dW) {
  if (param_propagate_down) {
    for (int i = 0; i < N; ++i) {
      /* sger: A += alpha x y^T */
      const Dtype alpha = top_diff[i];
      for (int r = 0; r < K1; ++r) {
        const Dtype ax = alpha * q[(size_t)i * K1 + r];
        for (int c = 0; c < K2; ++c)
          dW[(size_t)r * K2 + c] += ax * a[(size_t)i * K2 + c];
      }
    }
  }
  if (propagate_down0) {
    /* gemv NoTrans: dq_j = dT_j * W a_j */
    for (int j = 0; j < N; ++j)
      for (int r = 0; r < K1; ++r) {
        Dtype s = 0;
        for (int c = 0; c < K2; ++c)
          s += W[(size_t)r * K2 + c] * a[(size_t)j * K2 + c];
        dq[(size_t)j * K1 + r] = top_diff[j] * s;
      }
  }
  if (propagate_down1) {
    /* gemv Trans: da_j = dT_j * W^T q_j */
    for (int j = 0; j < N; ++j)
      for (int c = 0; c < K2; ++c) {
        Dtype s = 0;
        for (int r = 0; r < K1; ++r)
          s += W[(size_t)r * K2 + c] * q[(size_t)j * K1 + r];
        da[(size_t)j * K2 + c] = top_diff[j] * s;
      }
  }
}

/* ---------------------------------------------------------------------------
 * PairRankLoss.  Reference: pair_rank_loss_layer.cpp:26-52 (forward),
 * :55-84 (backward).  Bottoms a, b, y of `count` = N*C elements.
 *   forward (:28-37): ordered = a-b ; similar = ordered ; ordered *= y ;
 *     ordered = -1*ordered + 0*ordered (caffe_cpu_axpby with X == Y; with the
 *     reference's configured MKL cblas_saxpby this is the elementwise
 *     y := alpha*x + beta*y.  Under the ATLAS/OpenBLAS fallback of
 *     include/caffe/util/mkl_alternate.hpp:83-88 the aliased scal-then-axpy
 *     would zero the term; the author's build is MKL, which we follow);
 *     ordered += margin.
 *   loss (:40-50): sequential Dtype sum over i ascending of
 *     max(0,ordered) + |(1-y)*similar| ; divided by count.
 *   backward (:61-82): sign = (i==0 ? -1 : +1) * top_diff / count ;
 *     diff = sign*(1[ordered>0]*y - ((1-y)*similar>0 ? 1 : -1)*(1-y)),
 *     overwritten (not accumulated); strict '>' (the .cu uses '>=').
 * ordered/similar are the layer's cached blobs (outputs of forward).
 * ------------------------------------------------------------------------- */
void SFX(oracle_pairrank_forward)(int count, Dtype margin, const Dtype* a,
                                  const Dtype* b, const Dtype* y,
                                  Dtype* ordered, Dtype* similar, Dtype* loss) {
  for (int i = 0; i < count; ++i) ordered[i] = a[i] - b[i];
  for (int i = 0; i < count; ++i) similar[i] = ordered[i];
  for (int i = 0; i < count; ++i) ordered[i] = ordered[i] * y[i];
  for (int i = 0; i < count; ++i)
    ordered[i] = (Dtype)-1 * ordered[i] + (Dtype)0 * ordered[i];
  for (int i = 0; i < count; ++i) ordered[i] += margin;
  Dtype l = 0;
  for (int i = 0; i < count; ++i) {
    Dtype o = ordered[i] > (Dtype)0 ? ordered[i] : (Dtype)0; /* std::max(0,x) */
    l += o + FABS((1 - y[i]) * similar[i]);
  }
  l /= (Dtype)count;
  *loss = l;
}

void SFX(oracle_pairrank_backward)(int count, Dtype top_diff, const Dtype* y,
                                   const Dtype* ordered, const Dtype* similar,
                                   int propagate_down0, int propagate_down1,
                                   Dtype* da, Dtype* db) {
  for (int i = 0; i < 2; ++i) {
    if (!(i == 0 ? propagate_down0 : propagate_down1)) continue;
    Dtype sign = (i == 0) ? -1 : 1;
    Dtype* out = (i == 0) ? da : db;
    sign *= top_diff / count;
    for (int e = 0; e < count; ++e) {
      Dtype ordered_t = ordered[e] > 0 ? (Dtype)1 : (Dtype)0;
      Dtype similar_t = (1 - y[e]) * similar[e] > 0 ? (Dtype)1 : (Dtype)-1;
      out[e] = sign * (ordered_t * y[e] - similar_t * (1 - y[e]));
    }
  }
}

/* ---------------------------------------------------------------------------
 * RankAccuracy.  Reference: rank_accuracy_layer.cpp:36-50.
 * ------------------------------------------------------------------------- */
Dtype SFX(oracle_rank_accuracy)(int count, const Dtype* a, const Dtype* b,
                                const Dtype* label) {
  Dtype acc = 0;
  for (int i = 0; i < count; ++i)
    acc += (label[i] * (a[i] - b[i])) > 0 ? 1 : 0;
  return acc / count;
}

/* ---------------------------------------------------------------------------
 * Embed (the step before the path).
 * Reference: embed_layer.cpp:135-152 (forward), :155-180 (backward).
 *   forward : top[n] = weight[int(index[n])] (caffe_copy), then with a bias
 *             term  top = 1*(ones x bias) + 1*top  (gemm, K = 1).
 *   backward: weight_diff[int(index[n])] += top_diff[n] for n ASCENDING
 *             (caffe_axpy, alpha 1: y = 1*x + y), accumulating into the
 *             existing diff; bias_diff += top_diff^T ones (gemv).
 * ------------------------------------------------------------------------- */
void SFX(oracle_embed_forward)(int M, int N, const Dtype* index, const Dtype* weight,
                               const Dtype* bias, Dtype* top) {
  for (int n = 0; n < M; ++n) {
    const int idx = (int)index[n];
    memcpy(top + (size_t)n * N, weight + (size_t)idx * N, sizeof(Dtype) * N);
  }
  if (bias)
    for (int n = 0; n < M; ++n)
      for (int d = 0; d < N; ++d)
        top[(size_t)n * N + d] = (Dtype)1 * ((Dtype)1 * bias[d]) + (Dtype)1 * top[(size_t)n * N + d];
}

void SFX(oracle_embed_backward)(int M, int N, const Dtype* index, const Dtype* top_diff,
                                Dtype* weight_diff, Dtype* bias_diff) {
  for (int n = 0; n < M; ++n) {
    const int idx = (int)index[n];
    for (int d = 0; d < N; ++d)
      weight_diff[(size_t)idx * N + d] = (Dtype)1 * top_diff[(size_t)n * N + d] + weight_diff[(size_t)idx * N + d];
  }
  if (bias_diff)
    for (int d = 0; d < N; ++d) {
      Dtype s = 0;
      for (int n = 0; n < M; ++n) s += top_diff[(size_t)n * N + d] * (Dtype)1;
      bias_diff[d] = (Dtype)1 * s + (Dtype)1 * bias_diff[d];
    }
}
