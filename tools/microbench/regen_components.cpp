#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
typedef unsigned __int128 u128;
static inline uint64_t outf(u128 st){ uint64_t hi=(uint64_t)(st>>64), lo=(uint64_t)st; unsigned rot=(unsigned)(hi>>58); uint64_t x=hi^lo; return (x>>rot)|(x<<((64-rot)&63)); }
static uint32_t buf[65536];
int main(){
  const u128 A=((u128)0x2360ED051FC65DA4ull<<64)|0x4385DF649FCCF645ull; u128 c=((u128)12345<<64)|0x9999999999999999ull; c|=1;
  u128 s=42;
  auto t0=std::chrono::steady_clock::now();
  for(int rep=0;rep<100;rep++){ // serial gen 12288 steps
    for(int k=0;k<12288;k++){ s=s*A+c; uint64_t o=outf(s); buf[2*k]=(uint32_t)o; buf[2*k+1]=(uint32_t)(o>>32);} }
  auto t1=std::chrono::steady_clock::now();
  printf("serial gen: %.1f us per 12288 steps (%x)\n", std::chrono::duration<double,std::micro>(t1-t0).count()/100, buf[77]);
  const u128 A2=A*A,A3=A2*A,A4=A2*A2; const u128 C4=c*(A3+A2+A+1);
  t0=std::chrono::steady_clock::now();
  for(int rep=0;rep<100;rep++){ u128 x0=s*A+c,x1=x0*A+c,x2=x1*A+c,x3=x2*A+c; uint32_t*w=buf;
    for(int k=0;k<12288;k+=4){ uint64_t o0=outf(x0),o1=outf(x1),o2=outf(x2),o3=outf(x3);
      w[0]=(uint32_t)o0;w[1]=(uint32_t)(o0>>32);w[2]=(uint32_t)o1;w[3]=(uint32_t)(o1>>32);w[4]=(uint32_t)o2;w[5]=(uint32_t)(o2>>32);w[6]=(uint32_t)o3;w[7]=(uint32_t)(o3>>32);w+=8;
      x0=x0*A4+C4;x1=x1*A4+C4;x2=x2*A4+C4;x3=x3*A4+C4;} s=x3; }
  t1=std::chrono::steady_clock::now();
  printf("4-lane gen: %.1f us per 12288 steps (%x)\n", std::chrono::duration<double,std::micro>(t1-t0).count()/100, buf[77]);
  // scan
  static const uint8_t kMask[16]={0,1,3,3,7,7,7,7,15,15,15,15,15,15,15,15}; uint16_t acc[16];
  for(uint32_t low=0;low<16;low++){uint16_t m=0;for(int i=1;i<16;i++)m|=(uint16_t)(((low&kMask[i])<=(uint32_t)i)?1u:0u)<<i;acc[low]=m;}
  static uint8_t js[16384]; int pos=0;
  t0=std::chrono::steady_clock::now();
  for(int rep=0;rep<100;rep++){ pos=0; for(int row=0;row<1024;row++){uint8_t*jr=js+row*16;int i=15;const uint32_t*w=buf+pos;int u=0;while(i>=1){uint32_t low=w[u++]&15;jr[i]=low;i-=(acc[low]>>i)&1;}pos+=u;} }
  t1=std::chrono::steady_clock::now();
  printf("scan: %.1f us per regen, words=%d\n", std::chrono::duration<double,std::micro>(t1-t0).count()/100,pos);
  static uint8_t perm[16384]; for(int i=0;i<16384;i++)perm[i]=i&15;
  t0=std::chrono::steady_clock::now();
  for(int rep=0;rep<100;rep++){ for(int row=0;row<1024;row++){uint8_t*d=perm+row*16;const uint8_t*jr=js+row*16;
#pragma GCC unroll 15
    for(int i=15;i>=1;i--){uint32_t j=jr[i]&kMask[i];uint8_t a=d[i],b=d[j];d[j]=a;d[i]=b;}} }
  t1=std::chrono::steady_clock::now();
  printf("swap: %.1f us per regen (%d)\n", std::chrono::duration<double,std::micro>(t1-t0).count()/100,perm[5]);
}
