// Phase ablation of K1 (mash_sketch.hip): build once per PH_ABL value, run on the GPU box.
//   for a in 0 11 12 14 15; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPH_ABL=$a -I include -I poly_amd/csrc \
//       scripts/ubench/k1_ablate.hip poly_amd/csrc/runtime.hip poly_amd/csrc/multi_device.hip -o scripts/ubench/k1_ablate_$a; done
// (0..7: the tile pass's probes of round 1; 11..15: the slab pass's, round 5)
#include "../../poly_amd/csrc/mash_sketch.hip"
#include <cstdio>
#include <vector>
int main()
{
    const uint64_t n = 100000, L = 10000;
    const uint32_t k = 21, s = 1000;
    uint8_t *seqs; uint64_t *offs; uint32_t *out;
    hipMalloc(&seqs, n * L + 64); hipMalloc(&offs, (n + 1) * 8); hipMalloc(&out, n * s * 4);
    polyhip_synth_dna_dev(0xC2, 0, seqs, n * L, nullptr);
    std::vector<uint64_t> h(n + 1);
    for (uint64_t i = 0; i <= n; ++i) h[i] = i * L;
    hipMemcpy(offs, h.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    hipMemset(out, 0, n * s * 4);
    polyhip_mash_sketch_batch_dev(seqs, offs, n, k, s, out, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) polyhip_mash_sketch_batch_dev(seqs, offs, n, k, s, out, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    // checksum of all sketches: variants must agree with the baseline build word for word
    std::vector<uint32_t> ho(n * s);
    hipMemcpy(ho.data(), out, n * s * 4, hipMemcpyDeviceToHost);
    uint64_t sum = 0, x = 0;
    for (size_t i = 0; i < ho.size(); ++i) { sum += ho[i] * (uint64_t)(i % 1000003 + 1); x ^= ho[i]; }
    const k1::Launch PL = k1::plan(k, s);
    printf("checksum %016llx %08llx  smem_slab %zu capw %u capf_slab %u  ", (unsigned long long)sum, (unsigned long long)x, PL.smem_slab, PL.capw, PL.capf_slab);
    static const char *what[] = {"full kernel", "no premix", "1 chain block of 5", "no tail/fmix", "no select stores", "no bottom_s", "no global loads", "2 workgroups per CU",
                                 "", "", "", "slab: stage + premix + hash only", "slab: no bottom-s", "", "slab: per-read prologue + barriers only", "slab: no premix", "slab: 1 chain block of 5", "slab: fmix32 cut to one multiply"};
    const char *w = PH_ABL < (int)(sizeof what / sizeof what[0]) ? what[PH_ABL] : PH_ABL == 21 ? "slab: bottom-s without barriers" : PH_ABL == 22 ? "slab: bottom-s without its rank pass" : "";
    printf("PH_ABL=%d PH_BS_WIN=%d PH_SELBINS=%d %-40s %.3f ms per 100k reads\n", PH_ABL, PH_BS_WIN, PH_SELBINS, w, ms);
    return 0;
}
