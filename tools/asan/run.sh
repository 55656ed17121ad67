#!/bin/bash
# AddressSanitizer + UBSan over the host text side (CPU only; the GPU pool refuses sanitizer runs) on every golden
# input and fuzzed variants of them.  usage: bash tools/asan/run.sh
set -e
cd "$(dirname "$0")/../.."
OUT=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -pthread \
    -o $OUT/host_asan tools/asan/host_asan_main.cpp quasimodo_amd/csrc/qmvt_host.cpp -lz
FILES=$(find tests/golden -type f \( -name "*.vcf" -o -name "*.snps" \) | sort)
QM_HOST_THREADS=4 $OUT/host_asan $OUT $FILES
rm -rf $OUT
# the oracle's text functions (test infrastructure, but the checker deserves checking)
OUT=$(mktemp -d)
gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -o $OUT/oracle_asan tools/asan/oracle_asan_main.c oracle/qm_oracle.c -lm
$OUT/oracle_asan tests/golden/config1/input/nucmer/TA.maskrepeat.variants.vcf $(find tests/golden -type f -name "*.vcf" | sort)
$OUT/oracle_asan tests/golden/quirks/input/nucmer/r1_r2.maskrepeat.snps $(find tests/golden/quirks tests/golden/custom -type f -name "*.vcf" | sort)
rm -rf $OUT
# ThreadSanitizer: the multi-threaded scan (files >= 1 MB) and the shared allele dictionary
OUT=$(mktemp -d)
python3 - <<PY
import random
random.seed(3)
L = []
for i in range(60000):
    r = random.random()
    if r < 0.7: ref, alt = random.choice("ACGT"), random.choice("ACGT")
    elif r < 0.9: ref, alt = "A", "A" + "".join(random.choice("ACGT") for _ in range(random.randint(1, 8)))
    else: ref, alt = "".join(random.choice("ACGT") for _ in range(random.randint(14, 24))), "C"
    L.append("chr\t%d\t.\t%s\t%s\t%d\tPASS\tDP=%d;AF=0.1\n" % (i * 7 + 1, ref, alt, random.randint(0, 300), i))
open("$OUT/big.vcf", "w").write("##x\n#CHROM\n" + "".join(L))
PY
g++ -O1 -g -std=c++17 -fsanitize=thread -pthread -o $OUT/host_tsan tools/asan/host_asan_main.cpp quasimodo_amd/csrc/qmvt_host.cpp -lz
QM_HOST_THREADS=8 $OUT/host_tsan $OUT $OUT/big.vcf
rm -rf $OUT
