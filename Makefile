# Build everything for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
PKG      := flash-attention-cuda-c_amd
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     := gfx950
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++20 -fPIC -fno-slp-vectorize
LIB      := $(PKG)/libflash_attention.so
# the library's translation units: the C ABI + one unit per group of kernel instantiations (they compile in parallel)
KSRC     := $(wildcard $(PKG)/csrc/*.hip)
KOBJ     := $(patsubst $(PKG)/csrc/%.hip,build/obj/%.o,$(KSRC))
KHDR     := $(wildcard $(PKG)/csrc/*.h) $(PKG)/helpers.hpp include/flash_attention.h

# `all` = the product, its C++ harness and driver, the oracle, the microbenchmarks bench.py uses (about 1.5 min with -j4).
# `tune` = the kernel-variant A/B harness: ~70 kernel instantiations, 3 more minutes; not needed by tests or bench.
all: $(LIB) oracle $(PKG)/fa_main tests/fa_test tests/unit_kernels tests/micro/simd_mix tests/micro/valu_rates

tune: tests/fa_tune

lib: $(LIB)

# The d = 128 pair kernels run one 4-wave workgroup per CU (launch bounds 256, 1): hipcc would then place the MFMA accumulators in the
# accumulation registers and pay a v_accvgpr_read for every score the softmax touches (-6 ... -11 %, profiles/r04_tune_q_*); with the
# MFMAs in VGPR form the second half of the register file only takes what would otherwise spill
MFMA_VGPR := -mllvm -amdgpu-mfma-vgpr-form=1
build/obj/inst_bf16_pair_d128.o: HIPFLAGS += $(MFMA_VGPR)
tests/fa_tune tests/fa_tune_c128: HIPFLAGS += $(MFMA_VGPR)

build/obj/%.o: $(PKG)/csrc/%.hip $(KHDR)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(KOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(KOBJ)

oracle: oracle/liboracle_attention.so

# driver: the counterpart of the reference's main.cpp (device properties + the BASELINE configs sharded over the GPUs;
# RCCL reduces elapsed time and an output checksum; naive CPU attention + sampled check in the same run)
$(PKG)/fa_main: $(PKG)/main.cpp $(LIB) include/flash_attention.h
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++20 -o $@ $(PKG)/main.cpp -L$(PKG) -lflash_attention -Wl,-rpath,'$$ORIGIN' \
	    -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -lpthread

# test harness: the counterpart of the reference's tests/main.cu (launch + CPU check); links the oracle
oracle/liboracle_attention.so: oracle/cpu_attention.c oracle/cpu_attention.h
	$(MAKE) -s -C oracle

tests/fa_test: tests/main.cpp $(LIB) oracle/liboracle_attention.so
	$(HIPCC) -O2 -std=c++17 -o $@ tests/main.cpp -L$(PKG) -lflash_attention -Loracle -loracle_attention \
	    -Wl,-rpath,'$$ORIGIN/../$(PKG)' -Wl,-rpath,'$$ORIGIN/../oracle'

# unit tests of the MFMA fragment layouts and the LDS images (run by tests/conftest.py at session start, checked by tests/test_driver.py)
tests/unit_kernels: tests/unit_kernels.hip $(KHDR)
	$(HIPCC) $(HIPFLAGS) -o $@ tests/unit_kernels.hip

# kernel-variant A/B harness (tuning infrastructure)
tests/fa_tune: tests/fa_tune.hip $(KHDR) oracle/liboracle_attention.so
	$(HIPCC) $(HIPFLAGS) -o $@ tests/fa_tune.hip -Loracle -loracle_attention -Wl,-rpath,'$$ORIGIN/../oracle'

# ... with the causal d = 128 variants only (a third of the compile time)
tests/fa_tune_c128: tests/fa_tune.hip $(KHDR) oracle/liboracle_attention.so
	$(HIPCC) $(HIPFLAGS) -DFA_TUNE_CAUSAL_D128 -o $@ tests/fa_tune.hip -Loracle -loracle_attention -Wl,-rpath,'$$ORIGIN/../oracle'

# microbenchmarks: instruction issue rates; how busy 2 waves keep one SIMD's MFMA pipe; power-limited ceilings
tests/micro/%: tests/micro/%.hip
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++20 -o $@ $<

# Host-side sanitizers (there is no GPU ASan on this pool): the oracle under gcc's ASan + UBSan, and the library's
# HOST code -- argument validation, make_plan, snake_imbalance, plan_ex, shard_range: everything callable without a GPU -- under clang's
# (FlashAttention.hip rebuilt with -fsanitize=address,undefined; the kernel translation units are linked as built).  Each test file
# runs in a process of its own with the matching runtime preloaded; tests/main.cpp (the C++ harness) is compiled and linked under the
# same flags as a build check (running it needs a GPU).
ASAN_DIR  := build/asan
CLANG_ASAN_RT := $(shell /opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
GCC_ASAN_RT   := $(shell gcc -print-file-name=libasan.so)
SANFLAGS  := -fsanitize=address,undefined -fno-omit-frame-pointer -g
asan: $(LIB) oracle
	@mkdir -p $(ASAN_DIR)
	gcc -O1 -march=x86-64-v3 -fopenmp -fPIC -Wall -Wextra -std=c11 $(SANFLAGS) -shared -o $(ASAN_DIR)/liboracle_attention.so oracle/cpu_attention.c -lm
	$(HIPCC) $(HIPFLAGS) $(SANFLAGS) -fno-gpu-sanitize -shared-libsan -c -o $(ASAN_DIR)/FlashAttention.o $(PKG)/csrc/FlashAttention.hip
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared $(SANFLAGS) -shared-libsan -o $(ASAN_DIR)/libflash_attention.so $(ASAN_DIR)/FlashAttention.o $(filter-out build/obj/FlashAttention.o,$(KOBJ))
	$(HIPCC) -O1 -std=c++17 $(SANFLAGS) -shared-libsan -o $(ASAN_DIR)/fa_test tests/main.cpp -L$(ASAN_DIR) -lflash_attention -Loracle -loracle_attention \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../../oracle'
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 LD_PRELOAD=$(GCC_ASAN_RT) \
	    ORACLE_LIB_PATH=$(CURDIR)/$(ASAN_DIR)/liboracle_attention.so python -m pytest tests/test_oracle.py -q -p no:cacheprovider
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 LD_PRELOAD=$(CLANG_ASAN_RT) \
	    FA_LIB_PATH=$(CURDIR)/$(ASAN_DIR)/libflash_attention.so python -m pytest tests/test_abi.py tests/test_shard.py -q -p no:cacheprovider -m "not gpu"
	@echo "asan: oracle (gcc ASan+UBSan) and library host code (clang ASan+UBSan) clean"

asm: $(KSRC) $(KHDR)
	mkdir -p build && $(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/inst_bf16_d128.s $(PKG)/csrc/inst_bf16_d128.hip

clean:
	rm -f $(LIB) $(PKG)/fa_main tests/fa_test tests/fa_tune tests/unit_kernels tests/micro/simd_mix tests/micro/valu_rates tests/micro/atomic_latency oracle/liboracle_attention.so
	rm -rf build
.PHONY: all lib tune oracle clean asm asan
