"""Not applied: the row-tiled pair kernel of DESIGN 4b plus a verify mode (BSLV_VERIFY_TILED=1 runs it beside k_pair_flags_bits and
reports the first block whose flags differ).  Patch against bensolve_amd/csrc/poly_engine.hip; kept for the next attempt."""
p='/root/repo/bensolve_amd/csrc/poly_engine.hip'
s=open(p).read()
b=s.index("// emission for large facets: only the pair blocks that hold an adjacent pair")
tiled='''// The same for large facets, TI rows per workgroup (see DESIGN): LDS window of PB + PTI columns shared by PTI rows.
constexpr int PTI = 8;
__global__ __launch_bounds__(PB) void k_pair_flags_tiled(int d, const unsigned long long *bits, int nm, int W, unsigned char *pflag, Tri *bsum,
                                                          const int *fm_cnt, const int *fm_off, const int *fm_list, int *nzlist, int *nzcount)
{
    extern __shared__ unsigned long long s_t[];       // PTI x W row words | W x (PB + PTI) column words | 4 x W (one M per wave)
    const int i0 = blockIdx.y * PTI, c = blockIdx.x;
    const int jbase = i0 + 1 + c * PB;
    if (i0 >= nm - 1 || jbase >= nm) return;
    const int CW = PB + PTI;
    unsigned long long *rowsw = s_t, *colsw = s_t + (size_t)PTI * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *Mw = colsw + (size_t)W * CW + (size_t)wave * W;
    for (int x = threadIdx.x; x < PTI * W; x += PB) { const int r = x / W, w = x % W, i = i0 + r; rowsw[x] = i < nm ? bits[(size_t)w * nm + i] : 0ull; }
    for (int x = threadIdx.x; x < W * CW; x += PB) { const int w = x / CW, k = x % CW, j = jbase + k; colsw[x] = j < nm ? bits[(size_t)w * nm + j] : 0ull; }
    __syncthreads();
    const long long L1 = nm - 1, GL = pair_G(L1);
    for (int r = 0; r < PTI; r++) {
        const int i = i0 + r, j0 = i + 1 + c * PB;
        if (i >= nm - 1 || j0 >= nm) break;
        const unsigned long long *s_m = rowsw + (size_t)r * W;
        const int j = j0 + threadIdx.x, k = r + threadIdx.x;
        int nmut = 0;
        if (j < nm)
            for (int w = 0; w < W; w++) nmut += __popcll(s_m[w] & colsw[(size_t)w * CW + k]);
        bool cand = (j < nm) && ((d == 1) || (nmut >= d - 1));
        bool adj = cand;
        unsigned long long todo = __ballot(cand && d > 1);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int cj = j0 + (threadIdx.x - lane) + src;
            for (int w = lane; w < W; w += WAVE) Mw[w] = s_m[w] & bits[(size_t)w * nm + cj];
            __builtin_amdgcn_wave_barrier();
            bool found = false;
            {
                int bc = 0x7fffffff, bf = -1;
                for (int w = 0; w < W; w++) if ((Mw[w] >> lane) & 1ull) { const int cc = fm_cnt[w * 64 + lane]; if (cc < bc) { bc = cc; bf = w * 64 + lane; } }
                for (int o = 32; o > 0; o >>= 1) {
                    const int oc = __shfl_xor(bc, o, WAVE), of = __shfl_xor(bf, o, WAVE);
                    if (oc < bc || (oc == bc && of < bf)) { bc = oc; bf = of; }
                }
                const int *Lf = fm_list + fm_off[bf];
                for (int base = 0; base < bc; base += WAVE) {
                    bool hit = false;
                    if (base + lane < bc) {
                        const int wv = Lf[base + lane];
                        if (wv != i && wv != cj) {
                            hit = true;
                            for (int w = 0; w < W; w++)
                                if (Mw[w] & ~bits[(size_t)w * nm + wv]) { hit = false; break; }
                        }
                    }
                    if (__ballot(hit)) { found = true; break; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == src) adj = !found;
        }
        const long long vb = GL - pair_G(L1 - i) + c;
        pflag[(size_t)vb * PB + threadIdx.x] = adj ? 1 : 0;
        const int cnt = __syncthreads_count(adj);
        if (threadIdx.x == 0) {
            bsum[vb] = Tri{cnt, 0, 0};
            if (nzlist && cnt > 0) nzlist[atomicAdd(nzcount, 1)] = (int)vb;
        }
    }
}
__global__ void k_pair_verify(const unsigned char *pa, const unsigned char *pb2, const Tri *ba, const Tri *bb, long long nbp, long long *mism)
{
    const long long blk = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nbp) return;
    bool bad = ba[blk].a != bb[blk].a;
    for (int t = 0; t < PB && !bad; t++) bad = pa[blk * PB + t] != pb2[blk * PB + t];
    if (bad) { atomicMin((unsigned long long *)&mism[0], (unsigned long long)blk); atomicAdd((unsigned long long *)&mism[1], 1ull); }
}

'''
s=s[:b]+tiled+s[b:]
old_h='''        hipLaunchKernelGGL(k_pair_flags_bits, dim3((unsigned)nbp), dim3(PB), lds_bits, s, h->d, h->bits, nm, W, h->pflag, h->bsum,
                           fm ? (const int *)h->fm_cnt'''
new_h='''        const size_t lds_tiled = ((size_t)PTI * W + (size_t)W * (PB + PTI) + 4 * (size_t)W) * sizeof(unsigned long long);
        const int ngroups = (nm - 1 + PTI - 1) / PTI;
        if (fm && getenv("BSLV_VERIFY_TILED") && lds_tiled <= 48 * 1024 && ngroups <= 65535) {
            static unsigned char *pf2 = nullptr; static Tri *bs2 = nullptr; static size_t cap2 = 0; static long long *mism = nullptr;
            if ((size_t)nbp > cap2) { if (pf2) { (void)hipFree(pf2); (void)hipFree(bs2); } HIP_TRY(hipMalloc(&pf2, (size_t)nbp * PB)); HIP_TRY(hipMalloc(&bs2, ((size_t)nbp + 1) * sizeof(Tri))); cap2 = nbp; }
            if (!mism) HIP_TRY(hipMalloc(&mism, 16));
            long long init[2] = {0x7fffffffffffffffll, 0};
            HIP_TRY(hipMemcpyAsync(mism, init, 16, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_pair_flags_tiled, dim3((unsigned)((nm - 1 + PB - 1) / PB), (unsigned)ngroups), dim3(PB), lds_tiled, s, h->d, h->bits, nm, W, pf2, bs2,
                               (const int *)h->fm_cnt, (const int *)(h->fm_cnt + W * 64), (const int *)h->fm_list, (int *)nullptr, (int *)nullptr);
            hipLaunchKernelGGL(k_pair_flags_bits, dim3((unsigned)nbp), dim3(PB), lds_bits, s, h->d, h->bits, nm, W, h->pflag, h->bsum,
                               (const int *)h->fm_cnt, (const int *)(h->fm_cnt + W * 64), (const int *)h->fm_list, (int *)nullptr, (int *)nullptr);
            hipLaunchKernelGGL(k_pair_verify, dim3((unsigned)((nbp + 255) / 256)), dim3(256), 0, s, (const unsigned char *)h->pflag, (const unsigned char *)pf2, (const Tri *)h->bsum, (const Tri *)bs2, nbp, mism);
            long long res[2];
            HIP_TRY(hipMemcpyAsync(res, mism, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipGetLastError());
            if (res[1]) {
                // decode block -> (i, c)
                long long g = res[0], L = nm - 1, GL = pair_G(L); int lo = 0, hi = nm - 2;
                while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (GL - pair_G(L - mid) <= g) lo = mid; else hi = mid - 1; }
                fprintf(stderr, "VERIFY nm %d W %d nbp %lld: %lld mismatching blocks, first %lld = row %d chunk %lld (rows %d)\\n", nm, W, nbp, res[1], g, lo, g - (GL - pair_G(L - lo)), nm);
            } else fprintf(stderr, "VERIFY nm %d W %d nbp %lld ok\\n", nm, W, nbp);
        }
        hipLaunchKernelGGL(k_pair_flags_bits, dim3((unsigned)nbp), dim3(PB), lds_bits, s, h->d, h->bits, nm, W, h->pflag, h->bsum,
                           fm ? (const int *)h->fm_cnt'''
assert old_h in s
s=s.replace(old_h,new_h)
open(p,'w').write(s)
