// Design study (not product, not oracle): predictor -> replay -> verify scheme statistics.
// gcc -O2 -ffp-contract=off -o proto_v2 proto_v2.c -lm
#include "../oracle/surf96_oracle.c"
#include <stdio.h>


static long double secular_true(earth_t *E, double wvno, double omega)
{
    const int L = E->L;
    double e[5], ee[5], ca[5][5], v[10];
    long S=0;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double wvno2 = wvno * wvno;
    double xka = omega / E->a[L - 1];
    double xkb = omega / E->b[L - 1];
    double ra = sqrt((wvno + xka) * fabs(wvno - xka));
    double rb = sqrt((wvno + xkb) * fabs(wvno - xkb));
    double t = E->b[L - 1] / omega;
    double gammk = 2.0 * t * t;
    double gam = gammk * wvno2;
    double gamm1 = gam - 1.0;
    double rho1 = E->rho[L - 1];
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * ra * rb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - ra * rb;
    for (int m = L - 2; m >= 0; --m) {
        xka = omega / E->a[m];
        xkb = omega / E->b[m];
        t = E->b[m] / omega;
        gammk = 2.0 * t * t;
        gam = gammk * wvno2;
        ra = sqrt((wvno + xka) * fabs(wvno - xka));
        rb = sqrt((wvno + xkb) * fabs(wvno - xkb));
        double dpth = E->d[m];
        rho1 = E->rho[m];
        double p = ra * dpth;
        double q = rb * dpth;
        layer_products(p, q, ra, rb, wvno, xka, xkb, dpth, v);
        dunkin(ca, wvno2, gam, gammk, rho1, v);
        for (int i = 0; i < 5; ++i) {
            double cr = 0.0;
            for (int j = 0; j < 5; ++j) cr += e[j] * ca[j][i];
            ee[i] = cr;
        }
        double t1 = 0.0;
        for (int i = 0; i < 5; ++i) { double ai = fabs(ee[i]); if (ai > t1) t1 = ai; }
        int ex=0; if(t1>=1e-40){ frexp(t1,&ex);} 
        for (int i = 0; i < 5; ++i) e[i] = ldexp(ee[i], -ex);
        S+=ex;
        // the exponential factors pulled out by var(): exa = pex+sex is NOT in a0-scaled terms... keep as is
    }
    return ldexpl((long double)e[0], (int)S);
}

typedef struct { double lo, hi; int slo, shi; int iters; } bracket_t;

static int isgn(long double x){ return (x>0)-(x<0); }

static long long g_pred_evals=0, g_ties=0, g_mismatch=0, g_curves=0, g_points=0, g_maxiter=0, g_itersum=0, g_lanes=0;
static long long g_mismatch_curves=0, g_fail_curves=0;
static double TOLREL=1e-10; static int g_dbg=0;

// predictor for one period
static bracket_t predict(earth_t*E, double t, double g, int sigma0, double floorc, double top, double dc){
    bracket_t B; double omega=TWOPI/t;
    double x0=g; long double f0=secular_true(E,omega/x0,omega); g_pred_evals++;
    int it=1;
    int s0=isgn(f0);
    int dir=(s0*sigma0>=0)?+1:-1;
    double step=dc; int nscan=0;
    double a,b; long double fa,fb; int found=0;
    while(1){
        double x=x0+dir*step;
        int last=0;
        if(dir>0 && x>=top){x=top;last=1;}
        if(dir<0 && x<=floorc){x=floorc;last=1;}
        long double f=secular_true(E,omega/x,omega); g_pred_evals++; it++;
        if(isgn(f)!=s0){ if(dir>0){a=x0;fa=f0;b=x;fb=f;}else{a=x;fa=f;b=x0;fb=f0;} found=1;break;}
        x0=x;f0=f; nscan++; if(nscan>=2) step*=2; if(step>16*dc) step=16*dc;
        if(last)break;
    }
    if(!found){ // constant sign
        if(dir>0){B.lo=1e300;B.hi=1e301;B.slo=s0;B.shi=s0;} else {B.lo=-1e301;B.hi=-1e300;B.slo=s0;B.shi=s0;}
        B.iters=it; return B;
    }
    // Illinois
    int side=0; long double fa_w=fa, fb_w=fb;
    for(int n=0;n<40;n++){
        double tol=TOLREL*a;
        if(b-a<=tol)break;
        double x=(a*fb_w-b*fa_w)/(fb_w-fa_w);
        // keep strictly inside with margin; and guard step to collapse the stale side
        double margin=0.25*tol;
        if(!(x>a+margin)) x=a+ (b-a)*0.5*( (n>2)?1e-3:0.5);
        if(!(x<b-margin)) x=b- (b-a)*0.5*( (n>2)?1e-3:0.5);
        long double f=secular_true(E,omega/x,omega); g_pred_evals++; it++;
        if(g_dbg) printf("   n=%d a=%.15g b=%.15g w=%.3g x=%.15g f=%.3g fa=%.3g fb=%.3g\n",n,a,b,b-a,x,(double)f,(double)fa,(double)fb);
        if(isgn(f)==isgn(fa)){ a=x;fa=f;fa_w=f; if(side==-1) fb_w*=0.5; side=-1;}
        else { b=x;fb=f;fb_w=f; if(side==+1) fa_w*=0.5; side=+1;}
        // guard: if iterate converged one-sided, probe near
        if(n>=2){
            double w=b-a;
            // estimate: secant from true values
            double xs=(a*fb-b*fa)/(fb-fa);
            double da=xs-a, db=b-xs;
            double probe=-1;
            if(da<1e-3*w && db>4*tol) probe=xs+ (da>tol?2*da:2*tol);
            else if(db<1e-3*w && da>4*tol) probe=xs- (db>tol?2*db:2*tol);
            if(probe>a && probe<b){
                long double fp=secular_true(E,omega/probe,omega); g_pred_evals++; it++;
                if(isgn(fp)==isgn(fa)){a=probe;fa=fp;fa_w=fp;side=0;} else {b=probe;fb=fp;fb_w=fp;side=0;}
            }
        }
    }
    B.lo=a;B.hi=b;B.slo=isgn(fa);B.shi=isgn(fb);B.iters=it;
    return B;
}


static double HV=0.02, HD=0.004; static long long g_nsub_sum=0; static int g_use_homotopy=1;
static void gardner_table(int L,const double*d,const double*b,double*a,double*rho){ for(int i=0;i<L;i++){ a[i]=b[i]*1.75; rho[i]=(1741.0*pow(fabs(a[i]),0.25))/1000.0; } (void)d; }

// final tight bracket around x (known close to root) : returns bracket
static bracket_t tighten(earth_t*E,double t,double x,double slope_dx, double floorc,double top){
    bracket_t B; double omega=TWOPI/t; int it=0;
    long double f0=secular_true(E,omega/x,omega); g_pred_evals++; it++;
    // second point
    double h=slope_dx; if(fabs(h)<1e-7) h=1e-7;
    double x1=x+h; long double f1=secular_true(E,omega/x1,omega); g_pred_evals++; it++;
    double a,b; long double fa,fb; int have=0;
    for(int n=0;n<60;n++){
        if(isgn(f0)!=isgn(f1)){ if(x<x1){a=x;fa=f0;b=x1;fb=f1;}else{a=x1;fa=f1;b=x;fb=f0;} have=1;break;}
        // secant extrapolate, step at least 2x beyond
        long double den=f1-f0; double xs;
        if(den==0) xs=x1+(x1-x); else xs=(double)(x1 - f1*(x1-x)/den);
        double step=xs-x1; double mn=fabs(x1-x);
        // overshoot the secant estimate a bit to bracket
        step*=1.5; if(fabs(step)<mn*0.5) step=(step<0?-1:1)*mn*0.5; if(fabs(step)>0.02) step=(step<0?-1:1)*0.02;
        double xn=x1+step; if(xn<floorc)xn=floorc; if(xn>top)xn=top;
        if(xn==x1){break;}
        x=x1;f0=f1;x1=xn;f1=secular_true(E,omega/x1,omega); g_pred_evals++; it++;
    }
    if(!have){ B.lo=1e300;B.hi=1e301;B.slo=isgn(f1);B.shi=isgn(f1);B.iters=it; return B; }
    int side=0; long double fa_w=fa, fb_w=fb;
    for(int n=0;n<60;n++){
        double tol=TOLREL*a;
        if(b-a<=tol)break;
        double xx=(double)((a*fb_w-b*fa_w)/(fb_w-fa_w));
        double w=b-a;
        // if secant point hugs an end, push inside by a fraction so the bracket collapses
        double mgn=fmax(0.25*tol, 1e-3*w);
        if(!(xx>a+mgn)) xx=a+mgn;
        if(!(xx<b-mgn)) xx=b-mgn;
        long double f=secular_true(E,omega/xx,omega); g_pred_evals++; it++;
        if(isgn(f)==isgn(fa)){ a=xx;fa=f;fa_w=f; if(side==-1) fb_w*=0.5; side=-1;}
        else { b=xx;fb=f;fb_w=f; if(side==+1) fa_w*=0.5; side=+1;}
    }
    B.lo=a;B.hi=b;B.slo=isgn(fa);B.shi=isgn(fb);B.iters=it; return B;
}


// ---- Stage 1: coarse continuation with windows; returns number of rounds, fills coarse brackets (a,b,fa,fb)
static long long g_rev=0, g_s1_rounds=0, g_s1_giveup=0, g_s2_rounds=0;
static int g_policy=0; static long long g_rt[8]={0}; static long long g_res_hist[10]={0};
static int stage1(earth_t*E,const double*periods,int P,double cc,double betmx,double dc,const double*hint,
                  double*A,double*Bb,long double*FA,long double*FB,int*found){
    int rounds=0; int k0=0; double rprev=0; int sigma0=0; double last_inc=0; int have_inc=0;
    double carry_s=cc; int carry=0; // continuing an unresolved period
    long double carry_f=0; int carry_sign=0;
    while(k0<P){
        // choose g
        int g=4;
        if(g_policy==1 && k0>=2 && have_inc){ double inc=last_inc; if(inc<0)inc=0; if(8*inc+3<=8) g=8; else if(4*inc+3<=16) g=4; else if(2*inc+3<=32) g=2; else g=1; }
        if(g_policy==2 && k0>=2 && have_inc){ double inc=last_inc; if(inc<0)inc=0; if(8*inc+4<=8) g=8; else if(4*inc+5<=16) g=4; else if(2*inc+6<=32) g=2; else g=1; }
        if(hint){ g=8; while(g>1){ int kl=k0+g-1; if(kl>=P)kl=P-1; double span=(hint[kl]-hint[k0>0?k0-1:0])/dc; if(span<0)span=0; if(span+6<=64/g)break; g/=2; } }
        if(k0==0||carry) g=1;
        if(k0+g>P) g=P-k0;
        int W=64/g;
        double s = carry? carry_s : (k0==0? cc : rprev-1.5*dc);
        // evaluate
        static long double F[8][64];
        for(int jj=0;jj<g;jj++){ double om=TWOPI/periods[k0+jj]; for(int j=0;j<W;j++){ double x=s+j*dc; F[jj][j]=secular_true(E,om/x,om); g_pred_evals++; } }
        rounds++; if(carry) g_rt[1]++; else if(k0==0) g_rt[0]++; else g_rt[2+ (g==8?3:g==4?2:g==2?1:0)]++;
        if(k0==0&&!carry) sigma0=isgn(F[0][0]);
        int resolved=0;
        for(int jj=0;jj<g;jj++){
            int k=k0+jj;
            int start; int s1;
            if(carry&&jj==0){ start=-1; s1=carry_sign; }
            else if(k==0){ start=0; s1=isgn(F[0][0]); }
            else { double st=rprev-1.5*dc; start=(int)floor((st-s)/dc+1e-9); if(start<0)start=0; if(start>=W){ goto out;} s1=isgn(F[jj][start]); if(s1*sigma0<0){
                    if(resolved>0) goto out; /* handle in its own round */
                    /* reversed dispersion: scan downward from the start */
                    double st2=rprev-1.5*dc; double om=TWOPI/periods[k]; long double fprev=secular_true(E,om/st2,om); g_pred_evals++; int sp=isgn(fprev); double xp=st2; int gotd=0;
                    for(int j2=1;j2<64;j2++){ double x=st2-j2*dc; if(x<=cc){break;} long double f=secular_true(E,om/x,om); g_pred_evals++; if(isgn(f)!=sp){ A[k]=x;FA[k]=f;Bb[k]=xp;FB[k]=fprev;gotd=1;break;} xp=x;fprev=f; }
                    rounds++;
                    if(!gotd){ g_s1_giveup++; return -rounds; }
                    found[k]=1; rprev=0.5*(A[k]+Bb[k]); resolved++; carry=0; g_rev++;
                    goto out; } }
            int j; int got=0;
            for(j=start+1;j<W;j++){ if(isgn(F[jj][j])!=s1){ got=1;break;} if(s+j*dc>=betmx+dc){ found[k]=0; return rounds; } }
            if(!got){ // continue this period next round
                carry=1; carry_s=s+W*dc; carry_sign=s1; carry_f=F[jj][W-1]; A[k]=s+(W-1)*dc; FA[k]=F[jj][W-1]; k0=k; goto next; }
            if(j-1>=0){ A[k]=s+(j-1)*dc; FA[k]=F[jj][j-1]; } // else A,FA from carry
            Bb[k]=s+j*dc; FB[k]=F[jj][j]; found[k]=1; { double nr=0.5*(A[k]+Bb[k]); if(k>0){ last_inc=(nr-rprev)/dc; have_inc=1;} rprev=nr; } resolved++; carry=0;
        }
        out:
        g_res_hist[resolved<9?resolved:9]++;
        k0+=resolved; if(resolved==0 && !carry){ g_s1_giveup++; return -rounds; }
        next:;
        if(rounds>200){ g_s1_giveup++; return -rounds; }
    }
    return rounds;
}

// ---- Stage 1b: hint-anchored per-period windows
static int stage1b(earth_t*E,const double*periods,int P,double cc,double betmx,double dc,const double*hint,
                  double*A,double*Bb,long double*FA,long double*FB,int*found){
    int rounds=0; int k0=0; double rprev=0; int sigma0=0;
    int carry=0; double carry_base=0; int carry_sign=0; double carry_x=0; long double carry_f=0;
    int gmax=8;
    while(k0<P){
        int g=gmax; if(k0==0||carry) g=1; if(k0+g>P) g=P-k0;
        int W=64/g;
        static long double F[8][64]; double base[8];
        for(int jj=0;jj<g;jj++){
            int k=k0+jj;
            if(carry&&jj==0) base[jj]=carry_base;
            else if(k==0) base[jj]=cc;
            else { double pred_prev = rprev + (hint? (hint[k-1]-hint[k0-1]):0.0); /* predicted root of k-1 */
                   base[jj]= pred_prev-1.5*dc - (jj==0?0.0:(1.0+jj)*dc); if(jj==0) base[jj]=rprev-1.5*dc; }
            double om=TWOPI/periods[k]; for(int j=0;j<W;j++){ double x=base[jj]+j*dc; F[jj][j]=secular_true(E,om/x,om); g_pred_evals++; }
        }
        rounds++;
        if(k0==0&&!carry) sigma0=isgn(F[0][0]);
        int resolved=0;
        for(int jj=0;jj<g;jj++){
            int k=k0+jj; int start; int s1; double s=base[jj];
            if(carry&&jj==0){ start=-1; s1=carry_sign; }
            else if(k==0){ start=0; s1=isgn(F[0][0]); }
            else { double st=rprev-1.5*dc; double off=(st-s)/dc; if(off<-1e-9){ break; } start=(int)floor(off+1e-9); if(start>=W){ break; } s1=isgn(F[jj][start]);
                   if(s1*sigma0<0){
                    if(resolved>0) break;
                    double om=TWOPI/periods[k]; long double fprev=secular_true(E,om/st,om); g_pred_evals++; int sp=isgn(fprev); double xp=st; int gotd=0;
                    for(int j2=1;j2<64;j2++){ double x=st-j2*dc; if(x<=cc){break;} long double f=secular_true(E,om/x,om); g_pred_evals++; if(isgn(f)!=sp){ A[k]=x;FA[k]=f;Bb[k]=xp;FB[k]=fprev;gotd=1;break;} xp=x;fprev=f; }
                    rounds++;
                    if(!gotd){ g_s1_giveup++; return -rounds; }
                    found[k]=1; rprev=0.5*(A[k]+Bb[k]); resolved++; carry=0; g_rev++;
                    break; } }
            int j; int got=0;
            for(j=start+1;j<W;j++){ if(isgn(F[jj][j])!=s1){ got=1;break;} if(s+j*dc>=betmx+dc){ found[k]=0; return rounds; } }
            if(!got){ carry=1; carry_base=s+W*dc; carry_sign=s1; carry_x=s+(W-1)*dc; carry_f=F[jj][W-1]; if(W-1>start){A[k]=carry_x; FA[k]=carry_f;} resolved+=0; k0=k; goto next; }
            if(j-1>start || !(carry&&jj==0)){ if(j-1>=0){A[k]=s+(j-1)*dc; FA[k]=F[jj][j-1];} }
            Bb[k]=s+j*dc; FB[k]=F[jj][j]; found[k]=1; rprev=0.5*(A[k]+Bb[k]); resolved++; carry=0;
        }
        k0+=resolved;
        if(resolved<g/2 && gmax>1) gmax/=2; else if(resolved==g && gmax<8) gmax*=2;
        if(resolved==0){ gmax=1; }
        next:;
        if(rounds>300){ g_s1_giveup++; return -rounds; }
    }
    return rounds;
}


// ---- Stage 1c: per-period windows offset by the predicted increment
static int stage1c(earth_t*E,const double*periods,int P,double cc,double betmx,double dc,
                  double*A,double*Bb,long double*FA,long double*FB,int*found){
    int rounds=0; int k0=0; double rprev=0; int sigma0=0; double inc=1.0; int have_inc=0;
    int carry=0; double carry_base=0; int carry_sign=0;
    while(k0<P){
        int g;
        if(k0==0||carry) g=1;
        else { double ai=fabs(inc); if(!have_inc) g=4; else if(ai<=2.0) g=8; else if(ai<=5.0) g=4; else if(ai<=12) g=2; else g=1; }
        if(k0+g>P) g=P-k0;
        int W=64/g; if(g==5||g==6||g==7) W=8; if(g==3) W=16;
        static long double F[8][64]; double base[8];
        for(int jj=0;jj<g;jj++){
            int k=k0+jj;
            if(carry&&jj==0) base[jj]=carry_base;
            else if(k==0) base[jj]=cc;
            else { double pinc = have_inc? inc : 1.0; double pstart=rprev + jj*pinc*dc - 1.5*dc; int mlo = (jj==0)?0: (1+ (jj+1)/2); if(W>=16) mlo*=2; base[jj]=pstart - mlo*dc; if(base[jj]<cc-2*dc) base[jj]=cc-2*dc; }
            double om=TWOPI/periods[k]; for(int j=0;j<W;j++){ double x=base[jj]+j*dc; F[jj][j]=secular_true(E,om/x,om); g_pred_evals++; }
        }
        rounds++; if(carry) g_rt[1]++; else if(k0==0) g_rt[0]++; else g_rt[2+ (g>=8?3:g>=4?2:g>=2?1:0)]++;
        if(k0==0&&!carry) sigma0=isgn(F[0][0]);
        int resolved=0;
        for(int jj=0;jj<g;jj++){
            int k=k0+jj; int start; int s1; double s=base[jj];
            if(carry&&jj==0){ start=-1; s1=carry_sign; }
            else if(k==0){ start=0; s1=isgn(F[0][0]); }
            else { double st=rprev-1.5*dc; double off=(st-s)/dc; if(off<-1e-9) break; start=(int)floor(off+1e-9); if(start>=W-1) break; s1=isgn(F[jj][start]);
                   if(s1*sigma0<0){
                       // reversed: look downward inside the window
                       int j2; int gotd=0; for(j2=start-1;j2>=0;j2--){ if(isgn(F[jj][j2])!=s1){gotd=1;break;} }
                       if(gotd){ A[k]=s+j2*dc; FA[k]=F[jj][j2]; Bb[k]=s+(j2+1)*dc; FB[k]=F[jj][j2+1]; found[k]=1; double nr=0.5*(A[k]+Bb[k]); inc=(nr-rprev)/dc; have_inc=1; rprev=nr; resolved++; g_rev++; continue; }
                       if(resolved>0) break;
                       double om=TWOPI/periods[k]; long double fprev=secular_true(E,om/st,om); g_pred_evals++; int sp=isgn(fprev); double xp=st;
                       for(j2=1;j2<64;j2++){ double x=st-j2*dc; if(x<=cc){break;} long double f=secular_true(E,om/x,om); g_pred_evals++; if(isgn(f)!=sp){ A[k]=x;FA[k]=f;Bb[k]=xp;FB[k]=fprev;gotd=1;break;} xp=x;fprev=f; }
                       rounds++; g_rt[6]++;
                       if(!gotd){ g_s1_giveup++; return -rounds; }
                       found[k]=1; { double nr=0.5*(A[k]+Bb[k]); inc=(nr-rprev)/dc; have_inc=1; rprev=nr; } resolved++; g_rev++;
                       break; } }
            int j; int got=0;
            for(j=start+1;j<W;j++){ if(isgn(F[jj][j])!=s1){ got=1;break;} if(s+j*dc>=betmx+dc){ found[k]=0; return rounds; } }
            if(!got){ if(jj>0 && !(carry&&jj==0)) { /* let the next round restart this period with its own window */ break; }
                      carry=1; carry_base=s+W*dc; carry_sign=s1; A[k]=s+(W-1)*dc; FA[k]=F[jj][W-1]; k0=k; goto next; }
            if(j-1>=0){ A[k]=s+(j-1)*dc; FA[k]=F[jj][j-1]; }
            Bb[k]=s+j*dc; FB[k]=F[jj][j]; found[k]=1; { double nr=0.5*(A[k]+Bb[k]); if(k>0){ inc=(nr-rprev)/dc; have_inc=1;} rprev=nr; } resolved++; carry=0;
        }
        g_res_hist[resolved<9?resolved:9]++;
        k0+=resolved;
        if(resolved==0){ if(g==1 && !carry){ g_s1_giveup++; return -rounds; } inc = (fabs(inc)+1)*4; /* widen */ }
        next:;
        if(rounds>300){ g_s1_giveup++; return -rounds; }
    }
    return rounds;
}

static bracket_t refine_illinois(earth_t*E,double t,double a,double b,long double fa,long double fb,int*iters){
    bracket_t B; double omega=TWOPI/t; int it=0;
    int side=0; long double fa_w=fa, fb_w=fb;
    for(int n=0;n<60;n++){
        double tol=TOLREL*a;
        if(b-a<=tol)break;
        double xx=(double)((a*fb_w-b*fa_w)/(fb_w-fa_w));
        double w=b-a; double mgn=fmax(0.25*tol, 1e-3*w);
        if(!(xx>a+mgn)) xx=a+mgn;
        if(!(xx<b-mgn)) xx=b-mgn;
        long double f=secular_true(E,omega/xx,omega); g_pred_evals++; it++;
        if(isgn(f)==isgn(fa)){ a=xx;fa=f;fa_w=f; if(side==-1) fb_w*=0.5; side=-1;}
        else { b=xx;fb=f;fb_w=f; if(side==+1) fa_w*=0.5; side=+1;}
    }
    B.lo=a;B.hi=b;B.slo=isgn(fa);B.shi=isgn(fb);B.iters=it; *iters=it; return B;
}

// replay of getsol with inference; records points; returns ok, c
typedef struct { double x; int k; int s; } pt_t;
static pt_t pts[100000]; static int npts;

static int infer(earth_t*E,bracket_t*B,int k,double t,double x){
    int s;
    if(x<=B->lo) s=B->slo;
    else if(x>=B->hi) s=B->shi;
    else {
        g_ties++;
        double omega=TWOPI/t; double f=secular(E,omega/x,omega);
        s=isgn(f);
        if(s==B->slo) B->lo=x; else B->hi=x;
        return s; // real, not recorded
    }
    pts[npts].x=x;pts[npts].k=k;pts[npts].s=s;npts++;
    return s;
}

static int replay_period(earth_t*E, bracket_t*B, int k, double t1,double*c1io,double clow,double dc,double cm,double betmx,int ifirst,int*s1st){
    double c1=*c1io,c2;
    int s1=infer(E,B,k,t1,c1);
    if(ifirst)*s1st=s1;
    int idir=(ifirst|| (*s1st)*s1>=0)?+1:-1;
    for(;;){
        c2=(idir>0)?c1+dc:c1-dc;
        if(c2<=clow){idir=+1;c1=clow;continue;}
        int s2=infer(E,B,k,t1,c2);
        if(s1!=s2){
            double c3=0.5*(c1+c2);
            int s3=infer(E,B,k,t1,c3);
            for(int nctrl=2;nctrl<100;++nctrl){
                if(s3*s1<0){c2=c3;} else {c1=c3;s1=s3;}
                if(fabs(c1-c2)<=1e-6*c1)break;
                c3=0.5*(c1+c2);
                s3=infer(E,B,k,t1,c3);
            }
            *c1io=c3; return c3<=betmx;
        }
        c1=c2;s1=s2;
        if(c1<cm||c1>=betmx+dc)return 0;
    }
}

int main(int argc,char**argv){
    int nl=6,P=40; int NP=2*nl+1;
    int nchains=200, nsteps=200;
    if(argc>1)TOLREL=atof(argv[1]); if(argc>2)g_use_homotopy=atoi(argv[2]); if(argc>3)HV=atof(argv[3]); if(argc>4)HD=atof(argv[4]); if(argc>5)g_policy=atoi(argv[5]);
    double periods[128]; for(int k=0;k<P;k++){ double f=pow(10.0,0.3+(1.3-0.3)*(P-1-k)/(P-1)); periods[k]=1.0/f; }
    double lo[32],hi[32]; for(int i=0;i<nl;i++){lo[i]=0.001;hi[i]=0.15;} for(int i=nl;i<NP;i++){lo[i]=0.1;hi[i]=2.0;}
    srand48(1);
    double truem[13]={0.01,0.02,0.04,0.06,0.08,0.10,0.2,0.3,0.45,0.6,0.8,1.0,1.5};
    double obs[128]; orc_forward(truem,nl,1.75,periods,P,0,0,0,0.005,0.025,obs,NULL);
    for(int k=0;k<P;k++){ double u1=drand48(),u2=drand48(); obs[k]+=0.025*sqrt(-2*log(u1))*cos(6.2831853*u2);}
    double sig=0.025;
    long long lit_total=0; long long nvalid=0; long long hist_it[64]={0};
    for(int ic=0;ic<nchains;ic++){
        double m[32],cur[128]; int st;
        do{ for(int i=0;i<NP;i++)m[i]=lo[i]+(hi[i]-lo[i])*drand48(); orc_sort_layers(m,nl); st=orc_forward(m,nl,1.75,periods,P,0,0,0,0.005,0.025,cur,NULL);}while(st!=0);
        double logL=orc_loglik(obs,cur,P,0,&sig,1,NULL,0);
        for(int s=0;s<nsteps;s++){
            double test[32]; memcpy(test,m,sizeof(double)*NP);
            int kk=(int)(drand48()*NP); double cz=tan(3.14159265358979*(drand48()-0.5));
            double mn=(m[kk]-lo[kk])/(hi[kk]-lo[kk]); mn+=0.05*cz; test[kk]=mn*(hi[kk]-lo[kk])+lo[kk];
            orc_sort_layers(test,nl);
            int inb=1; for(int i=0;i<NP;i++) if(!(test[i]>=lo[i]&&test[i]<=hi[i]))inb=0;
            if(!inb)continue;
            nvalid++;
            // literal
            double lit[128]; long long ne=0; int lst=orc_forward(test,nl,1.75,periods,P,0,0,0,0.005,0.025,lit,&ne);
            lit_total+=ne;
            // v2
            double d[32],a[32],b[32],rho[32]; orc_layer_table(test,nl,1.75,d,a,b,rho);
            earth_t E={nl+1,d,a,b,rho,0,0,0};
            double betmx=-1e20,betmn=1e20;int jmn=0; for(int i=0;i<=nl;i++){ if(b[i]<betmn){betmn=b[i];jmn=i;} if(b[i]>betmx)betmx=b[i];}
            double cc=0.9*halfspace_rayleigh(a[jmn],b[jmn]); double dc=0.005;
            double om0=TWOPI/periods[0]; int sigma0=isgn(secular(&E,om0/cc,om0)); g_pred_evals++;
            bracket_t Bk[128]; int maxit=0;
            if(g_use_homotopy>=2){
                double A[128],Bb[128]; long double FA[128],FB[128]; int fnd[128]; for(int k=0;k<P;k++)fnd[k]=0;
                int r= g_use_homotopy==6 ? stage1c(&E,periods,P,cc,betmx,dc,A,Bb,FA,FB,fnd) : g_use_homotopy>=4 ? stage1b(&E,periods,P,cc,betmx,dc,g_use_homotopy==5?cur:NULL,A,Bb,FA,FB,fnd) : stage1(&E,periods,P,cc,betmx,dc,g_use_homotopy==3?cur:NULL,A,Bb,FA,FB,fnd);
                if(r<0){ /* give up -> v1 */ g_s1_rounds+=-r; g_curves++; g_points+=0; continue; }
                g_s1_rounds+=r; int mx2=0;
                for(int k=0;k<P;k++){ if(fnd[k]){ int it; Bk[k]=refine_illinois(&E,periods[k],A[k],Bb[k],FA[k],FB[k],&it); if(it>mx2)mx2=it; hist_it[it<63?it:63]++; g_lanes++; g_itersum+=it; } else { Bk[k].lo=1e300;Bk[k].hi=1e301;Bk[k].slo=Bk[k].shi=sigma0; } }
                g_s2_rounds+=mx2; maxit=r+mx2;
            } else if(g_use_homotopy==1){
            } else
            for(int k=0;k<P;k++){ double g=cur[k]-1.5*dc; if(g<cc)g=cc; if(g>betmx+dc) g=betmx; Bk[k]=predict(&E,periods[k],g,sigma0,cc-2*dc>1e-3?cc-2*dc:1e-3,betmx+dc,dc); if(Bk[k].iters>maxit)maxit=Bk[k].iters; hist_it[Bk[k].iters<63?Bk[k].iters:63]++; g_lanes++; g_itersum+=Bk[k].iters;}
            g_maxiter+=maxit;
            // replay
            npts=0; double c[128]; double c1=cc,clow=cc,cm=cc; int s1st=0; int ok=1; int kfail=-1;
            for(int k=0;k<P;k++){
                int ifirst=(k==0);
                if(ifirst){clow=cc;c1=cc;} else {clow=cm;c1=c[k-1]-1.5*dc;}
                ok=replay_period(&E,&Bk[k],k,periods[k],&c1,clow,dc,cm,betmx,ifirst,&s1st);
                if(!ok){kfail=k;break;}
                c[k]=c1;
            }
            // verify
            int mism=0;
            for(int i=0;i<npts;i++){ double om=TWOPI/periods[pts[i].k]; int s=isgn(secular(&E,om/pts[i].x,om)); if(s!=pts[i].s){ if(!mism && g_mismatch_curves<12){ int k=pts[i].k; printf("MISMATCH curve %lld k=%d x=%.12g inferred %d real %d  bracket lo=%.12g hi=%.12g slo=%d shi=%d cur[k]=%.6g lit[k]=%.6g lit[k-1]=%.6g cc=%.4g betmx=%.4g pt#%d of period (first pt idx?)\n",g_curves,k,pts[i].x,pts[i].s,s,Bk[k].lo,Bk[k].hi,Bk[k].slo,Bk[k].shi,cur[k],lst==0?lit[k]:-1, (lst==0&&k>0)?lit[k-1]:-1,cc,betmx,i);} mism++; } }
            g_points+=npts; g_curves++;
            int same = (ok==(lst==0));
            if(ok&&lst==0) for(int k=0;k<P;k++) if(c[k]!=lit[k]) same=0;
            if(mism){g_mismatch+=mism; g_mismatch_curves++;}
            if(!mism && !same){ printf("UNSOUND: verified but differs! chain %d step %d\n",ic,s); }
            if(lst!=0) g_fail_curves++;
            // MH accept with literal
            if(lst==0){ double lnew=orc_loglik(obs,lit,P,0,&sig,1,NULL,0); if(drand48()<=exp(lnew-logL)){ memcpy(m,test,sizeof(double)*NP); memcpy(cur,lit,sizeof(double)*P); logL=lnew; } }
        }
    }
    printf("tol %g curves %lld lit/curve %.1f pred_evals/curve %.1f avg_iters/lane %.2f avg_max_iters %.2f points/curve %.1f ties/curve %.4f mismatch_curves %lld (%.3f%%) noroot %lld\n",TOLREL,
        g_curves,(double)lit_total/g_curves,(double)g_pred_evals/g_curves,(double)g_itersum/g_lanes,(double)g_maxiter/g_curves,(double)g_points/g_curves,(double)g_ties/g_curves,g_mismatch_curves,100.0*g_mismatch_curves/g_curves,g_fail_curves);
    printf("round types: k0 %lld carry %lld g1 %lld g2 %lld g4 %lld g8 %lld | resolved hist:",g_rt[0],g_rt[1],g_rt[2],g_rt[3],g_rt[4],g_rt[5]); printf(" rev %lld ",g_rt[6]); for(int i=0;i<10;i++)printf(" %lld",g_res_hist[i]); printf("\n"); printf("reversed periods %lld ",g_rev); printf("avg nsub %.2f s1 rounds/curve %.2f s2 rounds/curve %.2f giveup %lld\n",(double)g_nsub_sum/g_curves,(double)g_s1_rounds/g_curves,(double)g_s2_rounds/g_curves,g_s1_giveup); printf("iters hist:"); for(int i=0;i<40;i++)printf(" %lld",hist_it[i]); printf("\n");
    return 0;
}
