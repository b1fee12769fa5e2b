// Cost of one monocular initialisation attempt (sst_two_view: 2 x 200 minimal-set models scored on every match, then the R|t hypotheses) on 1250 synthetic matches.
// Build: g++ -O2 -std=c++17 -pthread -I include -o /tmp/two_view_timing profiles/tools/two_view_timing.cpp send-slam_amd/csrc/ss_track.cpp
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "../../send-slam_amd/csrc/ss_track.h"
int main(){
    const int n=1250; std::mt19937 rng(3); std::uniform_real_distribution<double> ux(-4,4),uy(-2.5,2.5),uz(4,12); std::normal_distribution<double> noise(0,0.5);
    std::vector<double> x1(2*n),x2(2*n);
    for(int i=0;i<n;i++){double X=ux(rng),Y=uy(rng),Z=uz(rng); x1[2*i]=1000*X/Z+640+noise(rng); x1[2*i+1]=1000*Y/Z+360+noise(rng); x2[2*i]=1000*(X-0.3)/Z+640+noise(rng); x2[2*i+1]=1000*Y/Z+360+noise(rng);}
    sst_camera c{1000,1000,640,360,0,0,0,0};
    double R[9],t[3]; std::vector<uint8_t> tri; std::vector<double> p3d;
    for(int rep=0;rep<3;rep++){
        auto t0=std::chrono::steady_clock::now();
        int k=sst_two_view(c,n,x1.data(),x2.data(),R,t,tri,p3d);
        double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
        printf("two_view: %d triangulated, %.3f ms\n",k,ms);
    }
}
