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
LIB_HDRS   := $(wildcard include/*.h) $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inc)
OBJDIR     := $(LIBDIR)/obj
# The integrator's device code is compiled once per arithmetic mode (ptmi_device.hpp): strict, and `_da` = the
# reference's default OpenCL arithmetic (PTMI_FLAG_DEFAULT_ARITHMETIC).  Object files: `make -j` builds them side by side.
LIB_OBJS   := $(OBJDIR)/kernels.o $(OBJDIR)/kernel_wavefront.o $(OBJDIR)/kernels_da.o $(OBJDIR)/kernel_wavefront_da.o \
              $(OBJDIR)/display.o $(OBJDIR)/ptmi_api.o $(OBJDIR)/scene_layout.o $(OBJDIR)/bvh_build.o

.PHONY: all lib shim oracle ref clean resources
all: lib shim oracle

# C++ host shim with the reference's own backend signatures (namespace PathTracerNS) + the test driver that
# plays PathTracer_Main's part.  Plain g++: the shim is host code above the C ABI.
shim: $(LIBDIR)/libpathtracer_hip.so $(LIBDIR)/shim_driver
$(LIBDIR)/libpathtracer_hip.so: $(CSRC)/PathTracer_HIP.cpp include/pathtracer_backend.hpp include/ptmi.h $(LIBDIR)/libptmi.so
	g++ -std=c++14 -O2 -fPIC -shared -Iinclude $(CSRC)/PathTracer_HIP.cpp -o $@ -L$(LIBDIR) -lptmi -Wl,-rpath,'$$ORIGIN'
$(LIBDIR)/shim_driver: tests/shim_driver.cpp $(LIBDIR)/libpathtracer_hip.so
	g++ -std=c++14 -O2 -Iinclude tests/shim_driver.cpp -o $@ -L$(LIBDIR) -lpathtracer_hip -lptmi -Wl,-rpath,'$$ORIGIN'

lib:
	@$(MAKE) --no-print-directory -j4 $(LIBDIR)/libptmi.so
$(OBJDIR)/%_da.o: $(CSRC)/%.hip $(LIB_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DPTMI_DEFAULT_ARITHMETIC=1 -c $< -o $@
$(OBJDIR)/%.o: $(CSRC)/%.hip $(LIB_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OBJDIR)/%.o: $(CSRC)/%.cpp $(LIB_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIBDIR)/libptmi.so: $(LIB_OBJS)
	$(HIPCC) $(HIPFLAGS) -shared $(LIB_OBJS) -o $@

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf $(LIBDIR) oracle/build oracle/_ref

# register / LDS budget of both kernels (occupancy is VGPR-bound: read this after every kernel edit)
resources:
	@for f in kernels kernel_wavefront; do for m in 0 1; do $(HIPCC) $(HIPFLAGS) $(EXTRA) -DPTMI_DEFAULT_ARITHMETIC=$$m --cuda-device-only -c $(CSRC)/$$f.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo; done; done
