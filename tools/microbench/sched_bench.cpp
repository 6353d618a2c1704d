#include <chrono>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#define main_disabled
#include "../../ml2048_amd/csrc/hostrng.cpp"
int main(){
  ml2048_pcg64 r{123,456,789,1011|1,0,0};
  std::vector<uint8_t> perm(16384); for(int i=0;i<16384;i++)perm[i]=i&15; std::vector<float> flt(1024,0.5f);
  int32_t rs=0, ne=0; int n=128; std::vector<int32_t> sched(n*3); std::vector<uint8_t> ep(129*16384); std::vector<uint32_t> masks(129);
  for(int w=0;w<3;w++) ml2048_hostrng_schedule(&r,perm.data(),flt.data(),&rs,n,0.8,sched.data(),ep.data(),masks.data(),129,&ne);
  auto t0=std::chrono::steady_clock::now(); int tot=0;
  for(int rep=0;rep<100;rep++){ ml2048_hostrng_schedule(&r,perm.data(),flt.data(),&rs,n,0.8,sched.data(),ep.data(),masks.data(),129,&ne); tot+=ne-1;}
  auto t1=std::chrono::steady_clock::now();
  printf("%.3f ms per call, %.1f regens\n", std::chrono::duration<double,std::milli>(t1-t0).count()/100, tot/100.0);
}
