// shim_driver.cpp - plays PathTracer_Main's part (Controleur/PathTracer.cpp:47-96) for a scene dumped by the
// tests: fills a GlobalVars, then BVH_Create -> guards -> OpenCL_SetupContext -> OpenCL_InitializeMemory ->
// OpenCL_RunKernel, and writes imageColor / imageRayNb / histograms to a file.  Links against the shim only.
//
// scene file: u32 W,H,depth,sampler,nImages, nTri,nLights,nMat,nTex,nTexels; float camPos[4],camDir[4],camRight[4],
// camUp[4]; Sky; Triangle[nTri]; Light[]; Material[]; Texture[]; Uchar4[]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>

#include "pathtracer_backend.hpp"

using namespace PathTracerNS;

static int g_callbacks = 0;
static const GlobalVars* g_gv = nullptr;
static std::vector<uint64_t> g_shown;  // what the viewer saw at each callback: FNV-1a over imageColor, then over imageRayNb
static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static bool update_window()
{
    g_callbacks++;
    if (g_gv) {
        g_shown.push_back(fnv(g_gv->imageColor, sizeof(RGBAColor) * (size_t)g_gv->imageSize));
        g_shown.push_back(fnv(g_gv->imageRayNb, sizeof(float) * (size_t)g_gv->imageSize));
    }
    return true;
}

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return n == 0 || std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t h[10];
    if (!rd(f, h, 10)) return 2;
    GlobalVars gv;
    std::memset(&gv, 0, sizeof gv);
    gv.imageWidth = h[0]; gv.imageHeight = h[1]; gv.imageSize = h[0] * h[1]; gv.rayMaxDepth = h[2];
    const Sampler sampler = (Sampler)h[3];
    const uint nImages = h[4];
    gv.triangulationSize = h[5]; gv.lightsSize = h[6]; gv.materiauxSize = h[7]; gv.texturesSize = h[8];
    gv.texturesDataSize = h[9];
    std::vector<Triangle> tris(h[5]);
    std::vector<Light> lights(h[6] ? h[6] : 1);
    std::vector<Material> mats(h[7] ? h[7] : 1);
    std::vector<Texture> texs(h[8] ? h[8] : 1);
    std::vector<Uchar4> texels(h[9] ? h[9] : 1);
    if (!rd(f, &gv.cameraPosition, 1) || !rd(f, &gv.cameraDirection, 1) || !rd(f, &gv.cameraRight, 1) ||
        !rd(f, &gv.cameraUp, 1) || !rd(f, &gv.sky, 1) || !rd(f, tris.data(), h[5]) || !rd(f, lights.data(), h[6]) ||
        !rd(f, mats.data(), h[7]) || !rd(f, texs.data(), h[8]) || !rd(f, texels.data(), h[9]))
        return 2;
    std::fclose(f);
    gv.triangulation = tris.data(); gv.lights = lights.data(); gv.materiaux = mats.data();
    gv.textures = texs.data(); gv.texturesData = texels.data();
    std::vector<ptmi_float4> color(gv.imageSize);
    std::vector<float> count(gv.imageSize);
    std::vector<uint> depths(gv.rayMaxDepth + 1), bbx(PTMI_MAX_INTERSECTION_NUMBER), tri(PTMI_MAX_INTERSECTION_NUMBER);
    gv.imageColor = color.data(); gv.imageRayNb = count.data();
    gv.rayDepths = depths.data(); gv.rayIntersectedBBx = bbx.data(); gv.rayIntersectedTri = tri.data();
    double t1 = 0, t2 = 0, t3 = 0;
    g_gv = &gv;
    gv.printLogInfos = std::getenv("SHIM_DRIVER_LOG_INFO") != nullptr;  // the reference's -D LOG_INFO switch (OpenCL.cpp:310)
    try {
        BVH_Create(gv);
        if (gv.bvhMaxDepth >= PTMI_BVH_MAX_DEPTH || gv.lightsSize >= PTMI_MAX_LIGHT_SIZE) return 3;  // PathTracer.cpp:54-65
        OpenCL_SetupContext(gv, sampler);
        OpenCL_InitializeMemory(gv);
        OpenCL_RunKernel(gv, &update_window, nImages, &t1, &t2, &t3);
    } catch (std::exception const& e) {  // PathTracer.cpp:99-107
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 2;
    const uint32_t meta[3] = {(uint32_t)g_callbacks, gv.bvhSize, gv.bvhMaxDepth};
    std::fwrite(meta, 4, 3, o);
    std::fwrite(color.data(), 16, gv.imageSize, o);
    std::fwrite(count.data(), 4, gv.imageSize, o);
    std::fwrite(depths.data(), 4, depths.size(), o);
    std::fwrite(bbx.data(), 4, bbx.size(), o);
    std::fwrite(tri.data(), 4, tri.size(), o);
    const uint64_t n_shown = g_shown.size();  // trailer: the hashes of the images shown, in callback order
    std::fwrite(&n_shown, 8, 1, o);
    std::fwrite(g_shown.data(), 8, g_shown.size(), o);
    std::fclose(o);
    delete[] gv.bvh;
    return 0;
}
