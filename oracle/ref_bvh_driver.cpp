// ref_bvh_driver.cpp - TEST INFRASTRUCTURE, built only where /root/reference exists.
//
// Linked with the reference's own Controleur/PathTracer_BVH.cpp (compiled
// unmodified, see oracle/Makefile) into oracle/_ref/libref_bvh.so.  This file
// contributes no algorithm: it defines the `console` object that
// PathTracer_PreProc.h:79 declares extern, and a C entry point that fills the
// two GlobalVars fields BVH_Create reads and copies its result out.
#include <cstring>

#include "PathTracer_BVH.h"

mstream console;

extern "C" int ref_bvh_create(void* triangulation, unsigned n, void* nodes_out, unsigned nodes_capacity,
                              unsigned* bvh_size, unsigned* bvh_max_depth)
{
    PathTracerNS::GlobalVars gv;
    std::memset(&gv, 0, sizeof gv);
    gv.triangulation = static_cast<PathTracerNS::Triangle*>(triangulation);
    gv.triangulationSize = n;
    PathTracerNS::BVH_Create(gv);  // reorders `triangulation` in place, allocates gv.bvh
    int rc = 0;
    if (gv.bvhSize <= nodes_capacity) std::memcpy(nodes_out, gv.bvh, sizeof(PathTracerNS::Node) * gv.bvhSize);
    else rc = -1;
    *bvh_size = gv.bvhSize;
    *bvh_max_depth = gv.bvhMaxDepth;
    delete[] gv.bvh;
    return rc;
}

extern "C" unsigned ref_sizeof_node(void) { return sizeof(PathTracerNS::Node); }
extern "C" unsigned ref_sizeof_triangle(void) { return sizeof(PathTracerNS::Triangle); }
