# Builds the product library (HIP, gfx950 only) and the CPU oracle (test infrastructure).
#   make            -> diffusion-rs_amd/libflux_mi355x.so + diffusion-rs_amd/libflux_mi355x_alt.so + oracle/libflux_oracle.so
#   make lib / make alt / make oracle / make clean
# lib = the product library: the kernels the product runs.  alt = the TEST build of the same sources with -DFMI_ALT_KERNELS=1 (the superseded
# kernels compiled in as well, for the bit-identity cross-checks of tests/): two objects differ (attention.o, gemm_bf16.o), the rest is shared.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := diffusion-rs_amd
CSRC := $(PKG)/csrc
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value -Wno-unused-result -ffp-contract=on
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
LIB := $(PKG)/libflux_mi355x.so
ALT_LIB := $(PKG)/libflux_mi355x_alt.so
ALT_SRCS := attention gemm_bf16
ALT_OBJS := $(patsubst %,build/alt/%.o,$(ALT_SRCS)) $(filter-out $(patsubst %,build/%.o,$(ALT_SRCS)),$(OBJS))

# Source identity of the library: sha256 over csrc/*, include/*.h and this Makefile (sorted by path), first 16 hex digits.  It is compiled
# into the .so (fmi_build_id()), and __graft_entry__.build() recomputes it from the tree: a prebuilt binary that does not match the
# sources it travels with is rebuilt instead of passing the "does it build" check by being newer than them.
ID_FILES := $(sort $(wildcard $(CSRC)/*) $(wildcard include/*.h) Makefile)
BUILD_ID := $(shell cat $(ID_FILES) | sha256sum | cut -c1-16)

all: lib alt oracle
lib: $(LIB)
alt: $(ALT_LIB)
oracle:
	$(MAKE) -C oracle -s

HDRS := $(CSRC)/common.h $(CSRC)/gemm_w4q.h $(CSRC)/attention_w4.h $(CSRC)/attention_w4_loop.inc $(CSRC)/attention_w16.h $(CSRC)/attention_w16_loop.inc $(CSRC)/attention_w16f8_loop.inc $(CSRC)/attention_w32.h $(CSRC)/attention_w32_loop.inc $(CSRC)/attention_w16l.h $(CSRC)/attention_w16l_loop.inc $(CSRC)/attention_w16lf8_loop.inc $(CSRC)/attention_w16lf8pv_loop.inc include/flux_mi355x.h
build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -Ibuild -c $< -o $@

build/alt/%.o: $(CSRC)/%.hip $(HDRS) build/build_id.h
	@mkdir -p build/alt
	$(HIPCC) $(HIPFLAGS) -DFMI_ALT_KERNELS=1 -Ibuild -c $< -o $@

build/build_id.h: FORCE
	@mkdir -p build
	@echo '#define FMI_BUILD_ID "$(BUILD_ID)"' > $@.tmp
	@cmp -s $@.tmp $@ || cp $@.tmp $@
	@rm -f $@.tmp
# EVERY object depends on the id: it changes with any byte of any source, so a changed tree is rebuilt as a whole and the id that capi.o
# carries is the id of all the code that is linked (an object newer than an edited source can no longer ride along under a fresh id)
$(OBJS): build/build_id.h
FORCE:

# -Bsymbolic: calls between the library's own exported functions bind inside the library (the test build is loaded NEXT TO the product library)
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -Wl,-soname,libflux_mi355x.so -Wl,-Bsymbolic
$(ALT_LIB): $(ALT_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(ALT_OBJS) -Wl,-soname,libflux_mi355x_alt.so -Wl,-Bsymbolic

# the KV loop of attention_w4_kernel is generated assembly (committed; regenerate after editing the generator)
$(CSRC)/attention_w4_loop.inc: tools/gen_attention_w4_loop.py
	python3 tools/gen_attention_w4_loop.py > /dev/null

# attention_w16_kernel / attention_w32_kernel: the whole KV stream is generated (committed; regenerate after editing a generator)
$(CSRC)/attention_w16_loop.inc: tools/gen_attention_w16.py
	python3 tools/gen_attention_w16.py
$(CSRC)/attention_w16f8_loop.inc: tools/gen_attention_w16.py
	AW16_MODE=fp8qk python3 tools/gen_attention_w16.py
$(CSRC)/attention_w32_loop.inc: tools/gen_attention_w32.py
	python3 tools/gen_attention_w32.py
$(CSRC)/attention_w16l_loop.inc: tools/gen_attention_w16l.py
	python3 tools/gen_attention_w16l.py
$(CSRC)/attention_w16lf8_loop.inc: tools/gen_attention_w16l.py
	AW16L_MODE=fp8qk python3 tools/gen_attention_w16l.py
$(CSRC)/attention_w16lf8pv_loop.inc: tools/gen_attention_w16l.py
	AW16L_MODE=fp8pv python3 tools/gen_attention_w16l.py

clean:
	rm -rf build $(LIB) $(ALT_LIB)
	$(MAKE) -C oracle clean

.PHONY: all lib alt oracle clean FORCE

print-build-id:
	@echo $(BUILD_ID)
