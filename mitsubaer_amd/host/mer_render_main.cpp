// mer_render -- minimal driver over the C++ host mirror:  mer_render [-D key=value]... [-s spp] [-o out.npy] [--gpus N | --devices a,b,..] [--tiles] scene.xml
// (the reference's `mitsuba` CLI, src/mitsuba/mitsuba.cpp:154-246, reduced to what the hot path needs)
#include "mer_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv) {
    std::map<std::string, std::string> defines;
    std::string out = "out.npy", scenePath;
    int spp = 0, device = 0, layout = MER_LAYOUT_AUTO; unsigned long long seed = 0; bool raw = false;
    std::vector<int> devices; int shardMode = MER_SHARD_SAMPLES;          // several GPUs (the reference: -p <workers>, src/mitsuba/mitsuba.cpp:281)
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-D" && i + 1 < argc) { std::string kv = argv[++i]; size_t q = kv.find('='); if (q == std::string::npos) { std::fprintf(stderr, "-D expects key=value\n"); return 2; } defines[kv.substr(0, q)] = kv.substr(q + 1); }
        else if (a.rfind("-D", 0) == 0 && a.size() > 2) { std::string kv = a.substr(2); size_t q = kv.find('='); if (q != std::string::npos) defines[kv.substr(0, q)] = kv.substr(q + 1); }
        else if (a == "-o" && i + 1 < argc) out = argv[++i];
        else if (a == "-s" && i + 1 < argc) spp = std::atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], NULL, 10);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) { const int n = std::atoi(argv[++i]); if (n < 1) { std::fprintf(stderr, "--gpus expects a positive count\n"); return 2; } devices.clear(); for (int k = 0; k < n; k++) devices.push_back(k); }
        else if (a == "--devices" && i + 1 < argc) { devices.clear(); const std::string l = argv[++i]; size_t p = 0; while (p <= l.size()) { size_t e = l.find(',', p); if (e == std::string::npos) e = l.size(); if (e > p) devices.push_back(std::atoi(l.substr(p, e - p).c_str())); p = e + 1; } }
        else if (a == "--tiles") shardMode = MER_SHARD_TILES;
        else if (a == "--dense") layout = MER_LAYOUT_DENSE;
        else if (a == "--cell8") layout = MER_LAYOUT_CELL8;
        else if (a == "--raw") raw = true;
        else if (a == "-h" || a == "--help") { std::printf("usage: mer_render [-D key=value]... [-s spp] [-o out.npy|out.pfm|out.exr] [--raw] [--dense|--cell8] [--device n | --gpus N | --devices a,b,...] [--tiles] scene.xml\n"
                                                         "  --gpus N / --devices: one context per listed GPU, volumes replicated, samples (default) or 32x32 image tiles (--tiles) sharded over them,\n"
                                                         "  films reduced with RCCL (distinct devices) or peer copies (a device listed twice)\n"); return 0; }
        else scenePath = a;
    }
    if (scenePath.empty()) { std::fprintf(stderr, "mer_render: no scene file given\n"); return 2; }
    try {
        auto scene = merhost::loadScene(scenePath, defines);
        const int w = scene->sensor->film->width, h = scene->sensor->film->height;
        if (devices.empty()) devices.push_back(device);
        std::vector<float> film = scene->integrator->render(*scene, devices, shardMode, spp, seed, layout);
        const int frames = scene->sensor->film->frames();
        if (raw) merhost::writeNpy(out, film.data(), h, w, frames * 3 + 2);
        else {
            std::vector<float> rgb = merhost::develop(film, w, h, frames);
            const bool pfm = out.size() > 4 && out.substr(out.size() - 4) == ".pfm";
            const bool exr = out.size() > 4 && out.substr(out.size() - 4) == ".exr";
            if (frames == 1) { if (pfm) merhost::writePfm(out, rgb.data(), h, w); else if (exr) merhost::writeExr(out, rgb.data(), h, w); else merhost::writeNpy(out, rgb.data(), h, w, 3); }
            else if (exr) {
                for (int f = 0; f < frames; f++) {
                    char suffix[32]; std::snprintf(suffix, sizeof(suffix), "_%04d.exr", f);
                    merhost::writeExr(out.substr(0, out.size() - 4) + suffix, rgb.data() + (size_t) f * w * h * 3, h, w);
                }
            }
            else if (pfm) {                       // one file per frame, as the reference's hdrfilm writes <name>_<frame>
                for (int f = 0; f < frames; f++) {
                    char suffix[32]; std::snprintf(suffix, sizeof(suffix), "_%04d.pfm", f);
                    merhost::writePfm(out.substr(0, out.size() - 4) + suffix, rgb.data() + (size_t) f * w * h * 3, h, w);
                }
            } else merhost::writeNpy(out, rgb.data(), frames * h, w, 3);     // [frames*h][w][3]
        }
        std::printf("wrote %s (%dx%d)\n", out.c_str(), w, h);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "mer_render: %s\n", e.what());      // reference: the worker catches std::runtime_error and cancels
        return 1;
    }
    return 0;
}
