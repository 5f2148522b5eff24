#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds the unmodified reference (LongPhase-S) from the sources where they lie
# under /root/reference into oracle/_ref/ (git-ignored).  Nothing from the reference is copied into the repo:
# objects are compiled straight from /root/reference/src/**.cpp and the vendored htslib is built out-of-tree
# with VPATH.  htslib is configured zlib-only (no bz2/lzma/curl in this image); jemalloc is dropped (allocator
# only; the reference Makefile documents that variant).  Produces:
#   oracle/_ref/longphase-s-ref   the reference CLI (oracle + "reference" CPU baseline)
#   oracle/_ref/test_view         htslib's SAM->BAM(+BAI) helper used by the fixture generator
set -euo pipefail
REF=${LPS_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
O="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
  echo "build_ref: $REF not present (GPU box?) - keeping prebuilt oracle/_ref" >&2; exit 0
fi
mkdir -p "$O/hts/htscodecs/htscodecs" "$O/hts/cram" "$O/hts/test" "$O/obj"
H="$REF/htslib"
cat > "$O/hts/config.h" <<'CFG'
#ifndef _XOPEN_SOURCE
#define _XOPEN_SOURCE 600
#endif
#define HAVE_DRAND48 1
CFG
# version.h is what htslib's own Makefile rule writes: echo '#define HTS_VERSION_TEXT "$(PACKAGE_VERSION)"' with PACKAGE_VERSION := $(shell version.sh);
# the reference's own version.sh is run here, where it lies.  htscodecs ships its version.h in the tree when there is no git clone
# (Makefile:530-545); it is used from there when present.  config_vars.h holds compiler / flag strings for `htsfile --version` only.
echo "#define HTS_VERSION_TEXT \"$(sh "$H/version.sh")\"" > "$O/hts/version.h"
if [ ! -f "$H/htscodecs/htscodecs/version.h" ]; then echo '#define HTSCODECS_VERSION_TEXT "1.3.0"' > "$O/hts/htscodecs/htscodecs/version.h"; fi
cat > "$O/hts/config_vars.h" <<'CFG'
#define HTS_CC "gcc"
#define HTS_CPPFLAGS ""
#define HTS_CFLAGS "-O2"
#define HTS_LDFLAGS ""
#define HTS_LIBS "-lz -lm"
CFG
HTS_SRCS="kfunc.c kstring.c bcf_sr_sort.c bgzf.c errmod.c faidx.c header.c hfile.c hts.c hts_expr.c hts_os.c md5.c multipart.c probaln.c realn.c regidx.c region.c sam.c synced_bcf_reader.c vcf_sweep.c tbx.c textutils.c thread_pool.c vcf.c vcfutils.c
 cram/cram_codecs.c cram/cram_decode.c cram/cram_encode.c cram/cram_external.c cram/cram_index.c cram/cram_io.c cram/cram_stats.c cram/mFILE.c cram/open_trace_file.c cram/pooled_alloc.c cram/string_alloc.c
 htscodecs/htscodecs/arith_dynamic.c htscodecs/htscodecs/fqzcomp_qual.c htscodecs/htscodecs/htscodecs.c htscodecs/htscodecs/pack.c htscodecs/htscodecs/rANS_static4x16pr.c htscodecs/htscodecs/rANS_static32x16pr_avx2.c htscodecs/htscodecs/rANS_static32x16pr_avx512.c htscodecs/htscodecs/rANS_static32x16pr_sse4.c htscodecs/htscodecs/rANS_static32x16pr_neon.c htscodecs/htscodecs/rANS_static32x16pr.c htscodecs/htscodecs/rANS_static.c htscodecs/htscodecs/rle.c htscodecs/htscodecs/tokenise_name3.c htscodecs/htscodecs/utils.c"
pids=()
for s in $HTS_SRCS; do
  [ -f "$H/$s" ] || continue
  o="$O/hts/${s%.c}.o"
  if [ ! -f "$o" ] || [ "$H/$s" -nt "$o" ]; then
    gcc -O2 -fpic -I"$O/hts" -I"$H" -I"$H/htscodecs/htscodecs" -I"$O/hts/htscodecs/htscodecs" -c "$H/$s" -o "$o" &
    pids+=($!)
    if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
wait
rm -f "$O/hts/libhts.a"
ar rcs "$O/hts/libhts.a" $(find "$O/hts" -name '*.o' ! -path '*/test/*')
gcc -O2 -I"$O/hts" -I"$H" -o "$O/test_view" "$H/test/test_view.c" "$O/hts/libhts.a" -lz -lm -lpthread
pids=()
for f in "$REF"/src/*/*.cpp "$REF"/src/main.cpp; do
  o="$O/obj/$(basename "${f%.cpp}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then
    g++ -std=c++11 -O3 -fopenmp -w -I"$REF" -I"$REF/src" -I"$H" -c "$f" -o "$o" &
    pids+=($!)
    if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
wait
g++ -fopenmp -static-libstdc++ -static-libgcc -o "$O/longphase-s-ref" "$O"/obj/*.o "$O/hts/libhts.a" -lz -lm -lpthread
echo "build_ref: ok -> $O/longphase-s-ref, $O/test_view"
