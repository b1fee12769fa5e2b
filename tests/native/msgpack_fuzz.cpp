// Robustness check of frontdoor/ss_msgpack.h under ASan/UBSan: random and mutated payloads must
// either decode or throw, never read out of bounds.  Test infrastructure.
#include <cstdio>
#include <random>
#include <vector>
#include "../../send-slam_amd/frontdoor/ss_msgpack.h"

int main()
{
    std::mt19937 rng(7);
    long ok = 0, thrown = 0;
    // a valid pose packet as the mutation seed
    ssmp::packer pk;
    pk.pack_map(3);
    pk.pack("type"); pk.pack("frame");
    pk.pack("camera_id"); pk.pack(300);
    pk.pack("timestamp"); pk.pack(1.5);
    std::vector<uint8_t> seed = pk.buf;
    for (int it = 0; it < 200000; it++) {
        std::vector<uint8_t> b;
        if (it % 2) {
            b = seed;
            for (int k = 0; k < 1 + (int)(rng() % 4); k++) b[rng() % b.size()] = (uint8_t)rng();
            if (rng() % 3 == 0) b.resize(rng() % (b.size() + 1));
        } else {
            b.resize(rng() % 64);
            for (auto &x : b) x = (uint8_t)rng();
        }
        try {
            ssmp::value v = ssmp::decoder(b.data(), b.size()).parse();
            if (v.t == ssmp::type::MAP) {
                (void)v.find("type");
                for (auto &kv : v.map) { try { (void)kv.second.as_double(); } catch (...) {} try { (void)kv.second.as_int(); } catch (...) {} }
            }
            ok++;
        } catch (const std::exception &) {
            thrown++;
        }
    }
    std::printf("decoded=%ld rejected=%ld\n", ok, thrown);
    return 0;
}
