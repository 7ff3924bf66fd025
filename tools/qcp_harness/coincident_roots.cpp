#define __device__
#define __forceinline__ inline
#include <cstdio>
#include <cmath>
#include <random>
#include "ek_qcp.h"
static void randrot(std::mt19937_64 &rng, double R[9]) {
    std::normal_distribution<double> N(0,1);
    double q[4]; double n=0; for (int i=0;i<4;++i){q[i]=N(rng); n+=q[i]*q[i];} n=sqrt(n); for(int i=0;i<4;++i)q[i]/=n;
    double w=q[0],x=q[1],y=q[2],z=q[3];
    double M[9]={1-2*(y*y+z*z),2*(x*y-z*w),2*(x*z+y*w),2*(x*y+z*w),1-2*(x*x+z*z),2*(y*z-x*w),2*(x*z-y*w),2*(y*z+x*w),1-2*(x*x+y*y)};
    for(int i=0;i<9;++i)R[i]=M[i];
}
int main() {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0,1);
    double epss[] = {0, 1e-12, 1e-9, 1e-7, 1e-5, 1e-3};
    for (double eps : epss) {
        long bad=0, exits=0, total=0;
        for (long trial=0; trial<1500000; ++trial) {
            int A = 30;
            double s1 = A*(0.5+3*U(rng)), s2 = s1*U(rng), s3 = -s2*(1+eps);
            double Um[9], Vm[9]; randrot(rng,Um); randrot(rng,Vm);
            float S[9];
            double sg[3]={s1,s2,s3};
            for (int i=0;i<3;++i) for (int j=0;j<3;++j){ double v=0; for(int k=0;k<3;++k) v+=Um[3*i+k]*sg[k]*Vm[3*j+k]; S[3*i+j]=(float)v; }
            double lammax = s1 + s2 - fabs(s3) ;   // signed: s1+s2+s3
            lammax = s1 + s2 + s3;
            // alternative top root s1 - s2 - s3 = s1 + s2*eps (double-ish with the one above)
            double top = fmax(lammax, s1 - s2 - s3);
            double msd_t = pow(10.0, -6 + 6.5*U(rng));        // 1e-6 .. 3
            double Gsum = 2*top + A*msd_t;
            double Gx = Gsum*0.5, Gy = Gsum*0.5;
            float full = ek_rmsd_from_S(S, Gx, Gy, A);
            for (int r=0;r<4;++r) {
                float cur = (float)(full*(0.5+0.7*U(rng)));
                float b = ek_rmsd_from_S_below(S, Gx, Gy, A, cur);
                ++total;
                if (std::isinf(b)) { ++exits; if (full < cur) { ++bad; if (bad<4) printf("  eps=%g full=%.9g cur=%.9g msd_t=%g\n", eps, full, cur, msd_t);} }
                else if (b != full) { ++bad; printf(" differ\n"); }
            }
        }
        printf("eps=%g: %ld checks, %ld exits, %ld wrong\n", eps, total, exits, bad);
    }
}
