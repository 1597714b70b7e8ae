# Builds the product library (HIP, gfx950) and the test-side oracle.
#   make lib      -> opencl_pathtracer_amd/lib/libptmi.so      (hipcc, cross-compiles without a GPU)
#   make oracle   -> oracle/build/libpt_oracle.so              (gcc, CPU checker; test infrastructure)
#   make ref      -> oracle/_ref/*                             (only where /root/reference exists)
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := opencl_pathtracer_amd/csrc
LIBDIR     := opencl_pathtracer_amd/lib
# -ffp-contract=off: the numerics contract forbids fused multiply-add
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$(CSRC) -Wall -Wno-unused-function
LIB_SRCS   := $(CSRC)/kernels.hip $(CSRC)/kernel_wavefront.hip $(CSRC)/display.hip $(CSRC)/ptmi_api.cpp $(CSRC)/bvh_build.cpp
LIB_HDRS   := $(wildcard include/*.h) $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp)

.PHONY: all lib shim oracle ref clean resources
all: lib shim oracle

# C++ host shim with the reference's own backend signatures (namespace PathTracerNS) + the test driver that
# plays PathTracer_Main's part.  Plain g++: the shim is host code above the C ABI.
shim: $(LIBDIR)/libpathtracer_hip.so $(LIBDIR)/shim_driver
$(LIBDIR)/libpathtracer_hip.so: $(CSRC)/PathTracer_HIP.cpp include/pathtracer_backend.hpp include/ptmi.h $(LIBDIR)/libptmi.so
	g++ -std=c++14 -O2 -fPIC -shared -Iinclude $(CSRC)/PathTracer_HIP.cpp -o $@ -L$(LIBDIR) -lptmi -Wl,-rpath,'$$ORIGIN'
$(LIBDIR)/shim_driver: tests/shim_driver.cpp $(LIBDIR)/libpathtracer_hip.so
	g++ -std=c++14 -O2 -Iinclude tests/shim_driver.cpp -o $@ -L$(LIBDIR) -lpathtracer_hip -lptmi -Wl,-rpath,'$$ORIGIN'

lib: $(LIBDIR)/libptmi.so
$(LIBDIR)/libptmi.so: $(LIB_SRCS) $(LIB_HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared $(LIB_SRCS) -o $@

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf $(LIBDIR) oracle/build oracle/_ref

# register / LDS budget of both kernels (occupancy is VGPR-bound: read this after every kernel edit)
resources:
	@for f in kernels kernel_wavefront; do $(HIPCC) $(HIPFLAGS) $(EXTRA) --cuda-device-only -c $(CSRC)/$$f.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo; done
