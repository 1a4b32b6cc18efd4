/* oracle/mini.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * Restatement of the stencil helpers of src/MiniKernels.jl so that the golden values of
 * test/test_mini_kernels.jl can be checked against the same arithmetic the restated kernels use.
 * Interface is 1-based (i,j,k), exactly as the reference test calls them. */
#include "jrx_oracle.h"
#include "common.h"
#include <string.h>

/* 1-based accessors */
#define A2(i, j) A[IDX2(n1, (i)-1, (j)-1)]
#define A3(i, j, k) A[IDX3(n1, n2, (i)-1, (j)-1, (k)-1)]

/* MiniKernels.jl:208-234 : s = 0.0; for k, for j, for i: s += f(A[i,j,k]) */
double orc_mysum(int use_inv, const double *A, int n1, int n2, int n3,
                 int i0, int i1, int j0, int j1, int k0, int k1)
{
    (void)n3;
    double s = 0.0;
    for (int k = k0; k <= k1; k++)
        for (int j = j0; j <= j1; j++)
            for (int i = i0; i <= i1; i++) {
                double v = A3(i, j, k);
                s += use_inv ? inv(v) : v;
            }
    return s;
}

static double mysum2(int use_inv, const double *A, int n1, int i0, int i1, int j0, int j1)
{
    double s = 0.0;
    for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1; i++) {
            double v = A2(i, j);
            s += use_inv ? inv(v) : v;
        }
    return s;
}

double orc_mini2(const char *name, const double *A, int n1, int n2, double d, int i, int j)
{
#define IS(s) (strcmp(name, s) == 0)
    if (IS("center")) return A2(i, j);                    /* MiniKernels.jl:2-4   */
    if (IS("next")) return A2(i + 1, j + 1);              /* :5-6                 */
    if (IS("left")) return A2(i - 1, j);                  /* :7-8                 */
    if (IS("right")) return A2(i + 1, j);                 /* :9-10                */
    if (IS("back")) return A2(i, j - 1);                  /* :11-12               */
    if (IS("front")) return A2(i, j + 1);                 /* :13-14               */
    if (IS("_d_xa")) return (-A2(i, j) + A2(i + 1, j)) * d;           /* :37-39 */
    if (IS("_d_ya")) return (-A2(i, j) + A2(i, j + 1)) * d;           /* :40-42 */
    if (IS("_d_xi")) return (-A2(i, j + 1) + A2(i + 1, j + 1)) * d;   /* :46-48 */
    if (IS("_d_yi")) return (-A2(i + 1, j) + A2(i + 1, j + 1)) * d;   /* :49-51 */
    if (IS("_av")) return 0.25 * mysum2(0, A, n1, i + 1, i + 2, j + 1, j + 2);  /* :61-62 */
    if (IS("_av_a")) return 0.25 * mysum2(0, A, n1, i, i + 1, j, j + 1);        /* :63-64 */
    if (IS("_av_xa")) return (A2(i, j) + A2(i + 1, j)) * 0.5;                   /* :65-66 */
    if (IS("_av_ya")) return (A2(i, j) + A2(i, j + 1)) * 0.5;                   /* :67-68 */
    if (IS("_av_xi")) return (A2(i, j + 1) + A2(i + 1, j + 1)) * 0.5;           /* :69-70 */
    if (IS("_av_yi")) return (A2(i + 1, j) + A2(i + 1, j + 1)) * 0.5;           /* :71-72 */
    if (IS("_av_ai_clamped")) {                                                 /* :76-80 */
        int i0 = (int)clampi(i - 1, 1, n1), i1 = (int)clampi(i, 1, n1);
        int j0 = (int)clampi(j - 1, 1, n2), j1 = (int)clampi(j, 1, n2);
        return 0.25 * (A2(i0, j0) + A2(i1, j0) + A2(i0, j1) + A2(i1, j1));
    }
    if (IS("_harm")) return 4.0 * inv(mysum2(1, A, n1, i + 1, i + 2, j + 1, j + 2));  /* :83-85 */
    if (IS("_harm_a")) return 4.0 * inv(mysum2(1, A, n1, i, i + 1, j, j + 1));        /* :86-88 */
    if (IS("_harm_xa")) return 2.0 * inv(inv(A2(i + 1, j)) + inv(A2(i, j)));          /* :89-91 */
    if (IS("_harm_ya")) return 2.0 * inv(inv(A2(i, j + 1)) + inv(A2(i, j)));          /* :92-94 */
    return NAN;
}

double orc_div2(const double *Ax, const double *Ay, int n1, int n2, double _dx, double _dy, int i, int j)
{   /* MiniKernels.jl:57-58 */
    return orc_mini2("_d_xi", Ax, n1, n2, _dx, i, j) + orc_mini2("_d_yi", Ay, n1, n2, _dy, i, j);
}

double orc_mini3(const char *name, const double *A, int n1, int n2, int n3, double d, int i, int j, int k)
{
    if (IS("center") || IS("_current")) return A3(i, j, k);
    if (IS("next")) return A3(i + 1, j + 1, k + 1);
    if (IS("left")) return A3(i - 1, j, k);               /* :15-17 */
    if (IS("right")) return A3(i + 1, j, k);              /* :18-20 */
    if (IS("back")) return A3(i, j - 1, k);               /* :21-23 */
    if (IS("front")) return A3(i, j + 1, k);              /* :24-26 */
    if (IS("bot")) return A3(i, j, k - 1);                /* :27-29 */
    if (IS("top")) return A3(i, j, k + 1);                /* :30-32 */
    if (IS("_d_xa")) return (-A3(i, j, k) + A3(i + 1, j, k)) * d;
    if (IS("_d_ya")) return (-A3(i, j, k) + A3(i, j + 1, k)) * d;
    if (IS("_d_za")) return (-A3(i, j, k) + A3(i, j, k + 1)) * d;                   /* :43-45 */
    if (IS("_d_xi")) return (-A3(i, j + 1, k + 1) + A3(i + 1, j + 1, k + 1)) * d;   /* :53 */
    if (IS("_d_yi")) return (-A3(i + 1, j, k + 1) + A3(i + 1, j + 1, k + 1)) * d;   /* :54 */
    if (IS("_d_zi")) return (-A3(i + 1, j + 1, k) + A3(i + 1, j + 1, k + 1)) * d;   /* :55 */
#define MS(f, a, b, c, e, g, h) orc_mysum(f, A, n1, n2, n3, a, b, c, e, g, h)
    if (IS("_av")) return 0.125 * MS(0, i, i + 1, j, j + 1, k, k + 1);   /* :108-109 */
    if (IS("_av_x")) return 0.5 * (A3(i, j, k) + A3(i + 1, j, k));       /* :110-111 */
    if (IS("_av_y")) return 0.5 * (A3(i, j, k) + A3(i, j + 1, k));
    if (IS("_av_z")) return 0.5 * (A3(i, j, k) + A3(i, j, k + 1));
    if (IS("_av_xy")) return 0.25 * MS(0, i, i + 1, j, j + 1, k, k);     /* :116-117 */
    if (IS("_av_xz")) return 0.25 * MS(0, i, i + 1, j, j, k, k + 1);
    if (IS("_av_yz")) return 0.25 * MS(0, i, i, j, j + 1, k, k + 1);
    if (IS("_av_xyi")) return 0.25 * MS(0, i - 1, i, j - 1, j, k, k);    /* :122-123 */
    if (IS("_av_xzi")) return 0.25 * MS(0, i - 1, i, j, j, k - 1, k);
    if (IS("_av_yzi")) return 0.25 * MS(0, i, i, j - 1, j, k - 1, k);
    if (IS("_harm_x")) return 2.0 * inv(inv(A3(i, j, k)) + inv(A3(i + 1, j, k)));   /* :149-151 */
    if (IS("_harm_y")) return 2.0 * inv(inv(A3(i, j, k)) + inv(A3(i, j + 1, k)));
    if (IS("_harm_z")) return 2.0 * inv(inv(A3(i, j, k)) + inv(A3(i, j, k + 1)));
    if (IS("_harm_xy")) return 4.0 * inv(MS(1, i, i + 1, j, j + 1, k, k));
    if (IS("_harm_xz")) return 4.0 * inv(MS(1, i, i + 1, j, j, k, k + 1));
    if (IS("_harm_yz")) return 4.0 * inv(MS(1, i, i, j, j + 1, k, k + 1));
    if (IS("_harm_xyi")) return 4.0 * inv(MS(1, i - 1, i, j - 1, j, k, k));
    if (IS("_harm_xzi")) return 4.0 * inv(MS(1, i - 1, i, j, j, k - 1, k));
    if (IS("_harm_yzi")) return 4.0 * inv(MS(1, i, i, j - 1, j, k - 1, k));
    int clamped_av = IS("_av_xyi_clamped") || IS("_av_xzi_clamped") || IS("_av_yzi_clamped");
    int clamped_hm = IS("_harm_xyi_clamped") || IS("_harm_xzi_clamped") || IS("_harm_yzi_clamped");
    if (clamped_av || clamped_hm) {                                      /* :133-147, :180-194 */
        int i0 = (int)clampi(i - 1, 1, n1), i1 = (int)clampi(i, 1, n1);
        int j0 = (int)clampi(j - 1, 1, n2), j1 = (int)clampi(j, 1, n2);
        int k0 = (int)clampi(k - 1, 1, n3), k1 = (int)clampi(k, 1, n3);
        double a, b, c, e;
        if (strstr(name, "_xyi")) { a = A3(i0, j0, k); b = A3(i1, j0, k); c = A3(i0, j1, k); e = A3(i1, j1, k); }
        else if (strstr(name, "_xzi")) { a = A3(i0, j, k0); b = A3(i1, j, k0); c = A3(i0, j, k1); e = A3(i1, j, k1); }
        else { a = A3(i, j0, k0); b = A3(i, j1, k0); c = A3(i, j0, k1); e = A3(i, j1, k1); }
        if (clamped_av) return 0.25 * (a + b + c + e);
        return 4.0 * inv(inv(a) + inv(b) + inv(c) + inv(e));
    }
    return NAN;
}

double orc_div3(const double *Ax, const double *Ay, const double *Az, int n1, int n2, int n3,
                double _dx, double _dy, double _dz, int i, int j, int k)
{   /* MiniKernels.jl:104-105 */
    return orc_mini3("_d_xi", Ax, n1, n2, n3, _dx, i, j, k) + orc_mini3("_d_yi", Ay, n1, n2, n3, _dy, i, j, k) +
           orc_mini3("_d_zi", Az, n1, n2, n3, _dz, i, j, k);
}

double orc_compute_dtau_r(double theta_dtau, double eta, double _Gdt) { return compute_dtau_r(theta_dtau, eta, _Gdt); }
