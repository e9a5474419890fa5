#include "../oracle/surf96_oracle.c"
#include <stdio.h>
int main(){
    int nl=6,P=40,NP=13; double periods[128]; for(int k=0;k<P;k++){ double f=pow(10.0,0.3+(1.3-0.3)*(P-1-k)/(P-1)); periods[k]=1.0/f; }
    double lo[32],hi[32]; for(int i=0;i<nl;i++){lo[i]=0.001;hi[i]=0.15;} for(int i=nl;i<NP;i++){lo[i]=0.1;hi[i]=2.0;}
    srand48(1);
    double truem[13]={0.01,0.02,0.04,0.06,0.08,0.10,0.2,0.3,0.45,0.6,0.8,1.0,1.5};
    double obs[128]; orc_forward(truem,nl,1.75,periods,P,0,0,0,0.005,0.025,obs,NULL);
    for(int k=0;k<P;k++){ double u1=drand48(),u2=drand48(); obs[k]+=0.025*sqrt(-2*log(u1))*cos(6.2831853*u2);}
    double sig=0.025; double thr[6]={1,2,5,10,20,50}; long long cnt[2][6]={{0}}, n[2]={0}, nacc[2]={0}, scan0[2]={0}, nfail[2]={0};
    int nchains=100,nsteps=3000;
    for(int ic=0;ic<nchains;ic++){
        double m[32],cur[128]; int st;
        do{ for(int i=0;i<NP;i++)m[i]=lo[i]+(hi[i]-lo[i])*drand48(); orc_sort_layers(m,nl); st=orc_forward(m,nl,1.75,periods,P,0,0,0,0.005,0.025,cur,NULL);}while(st!=0);
        double logL=orc_loglik(obs,cur,P,0,&sig,1,NULL,0);
        for(int s=0;s<nsteps;s++){
            int ph = s>=2200;  // phase 0: the bench window (first 2200 steps), phase 1: later
            double test[32]; memcpy(test,m,sizeof(double)*NP);
            int kk=(int)(drand48()*NP); double cz=tan(3.14159265358979*(drand48()-0.5));
            double mn=(m[kk]-lo[kk])/(hi[kk]-lo[kk]); mn+=0.05*cz; test[kk]=mn*(hi[kk]-lo[kk])+lo[kk];
            orc_sort_layers(test,nl);
            int inb=1; for(int i=0;i<NP;i++) if(!(test[i]>=lo[i]&&test[i]<=hi[i]))inb=0;
            if(!inb)continue;
            double lit[128]; int lst=orc_forward(test,nl,1.75,periods,P,0,0,0,0.005,0.025,lit,NULL);
            if(lst!=0){nfail[ph]++;continue;}
            double mx=0; for(int k=0;k<P;k++){ double d=fabs(lit[k]-cur[k])/0.005; if(d>mx)mx=d;}
            n[ph]++; for(int j=0;j<6;j++) if(mx>thr[j])cnt[ph][j]++;
            double lnew=orc_loglik(obs,lit,P,0,&sig,1,NULL,0); if(drand48()<=exp(lnew-logL)){ memcpy(m,test,sizeof(double)*NP); memcpy(cur,lit,sizeof(double)*P); logL=lnew; nacc[ph]++; }
        }
        if(ic%20==0) fprintf(stderr,"chain %d logL %.1f\n",ic,logL);
    }
    for(int ph=0;ph<2;ph++){ printf("phase %d: valid %lld acc %.3f noroot %lld | frac max|shift|>",ph,n[ph],(double)nacc[ph]/n[ph],nfail[ph]); for(int j=0;j<6;j++)printf(" %gdc:%.3f",thr[j],(double)cnt[ph][j]/n[ph]); printf("\n"); }
}
