// NOPE_F16X2 activation-range tracking, shared by the network runtimes (unet_runtime.hip, ldm_runtime.hip).
//
// The f16 + MX-fp8 tile forms its A operands from a' = a * 2^-t (t per layer: word 3 of the tail of the layer's second weight pack,
// nope_common.h: kX2*): f16(a') saturates at 65504, e4m3(a' * 2^-2) at |a'| = 1792 and runs out of significant bits below 2^-4.  A launch
// whose LARGEST |a'| lies outside [2^-4, 1792] computed its cross terms from saturated / subnormal operands -- plain-f16 accuracy instead
// of ~2^-15 per product.  This file keeps every forward inside the window without putting a host synchronisation into the step:
//   * max |a| comes from the PRODUCERS of the tensors a layer reads.  Every tensor a forward writes that a two-pass launch consumes owns
//     a range slot (kX2SlotWords device words, amax_publish): filled by gn_apply in passing (spare VALU slots of a kernel that waits
//     for HBM), by the wide conv epilogue for the few conv-produced tensors that go straight into a conv (ConvArgs::out_amax), or by an
//     absmax pass when neither applies.  (Tracking inside the conv kernels: +5 % of the 512-template step, profiles/r06c_*; one word
//     per tensor: 24 576 same-address atomics per level-0 gn_apply, +0.3 ms per launch -- hence slots of 32 lines.)
//   * X2Fwd is a forward's bookkeeping: which slot holds the maximum of the tensor at a buffer address, which slots each layer's
//     two-pass launches of THIS forward read (a per-forward table: small banks leave most layers on the three-pass kernels).
//   * finish(): behind the forward's last kernel, x2_verdict_kernel folds the slots, judges every layer against its window and writes
//     the verdict to mapped host memory; x2_poison_kernel overwrites the output of an out-of-range forward with NaNs -- inaccurate
//     values never pass for accurate ones.
//   * X2Range::poll(): the host reads the verdicts that have arrived (no waiting), re-centres the shifts of flagged layers (max |a'| in
//     [256, 512)) and reports; the new shifts travel on the caller's stream.  The runtimes poll at the start of every forward.
#pragma once
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "nope_common.h"

namespace nope {

struct X2Range {
    static constexpr int SLOTS = 512, RING = 16;
    std::vector<int*> tails;                 // per layer: device pointer to the pack's 16-byte tail
    std::vector<int> t;                      // host copy of the current shifts
    std::vector<char> moved;                 // has the layer's shift ever left 0? (only never-moved layers take the t == 0 kernel instantiation)
    unsigned* amax = nullptr;                // device, SLOTS range slots of kX2SlotWords words
    int* tab_dev = nullptr;                  // device [n][5]: {t, four slots} of the forward being judged
    int* tab_pin = nullptr;                  // pinned host ring of RING such tables (one per forward in flight)
    int* t_pin = nullptr;                    // pinned host [n]: staging of re-centred shifts on their way into the packs' tails
    unsigned* status_dev = nullptr;          // device [4]: see x2_verdict_kernel
    unsigned* host = nullptr;                // mapped host [4 + 2 n]
    unsigned* host_dev = nullptr;            // ... its device address
    bool off = false;                        // every launch as NOPE_BF16X3 (the fallback for non-finite activations)
    unsigned ring_i = 0, seen_serial = 0, seen_bad = 0, seen_inf = 0;
    std::mutex mu;

    int add_layer(void* w_x2, size_t pack_bytes) {
        tails.push_back(reinterpret_cast<int*>((unsigned char*)w_x2 + pack_bytes - kX2TailBytes));
        return (int)tails.size() - 1;
    }
    bool active() const { return !tails.empty() && amax && host && !off; }
    bool t_zero(int layer) { std::lock_guard<std::mutex> lock(mu); return !moved[layer]; }

    // after the last add_layer: device / pinned buffers (dmalloc: the owner's allocator for device memory)
    template <class Alloc> int init(Alloc&& dmalloc, hipStream_t s) {
        const size_t n = tails.size();
        if (!n) return NOPE_OK;
        amax = (unsigned*)dmalloc((size_t)SLOTS * kX2SlotWords * sizeof(unsigned));
        tab_dev = (int*)dmalloc(n * 5 * sizeof(int));
        status_dev = (unsigned*)dmalloc(4 * sizeof(unsigned));
        if (!amax || !tab_dev || !status_dev) return NOPE_ERR_ALLOC;
        if (hipMemsetAsync(amax, 0, (size_t)SLOTS * kX2SlotWords * sizeof(unsigned), s) != hipSuccess) return NOPE_ERR_LAUNCH;
        if (hipMemsetAsync(status_dev, 0, 4 * sizeof(unsigned), s) != hipSuccess) return NOPE_ERR_LAUNCH;
        const size_t hostw = 4 + 2 * n;
        if (hipHostMalloc((void**)&tab_pin, (size_t)RING * n * 5 * sizeof(int), 0) != hipSuccess ||
            hipHostMalloc((void**)&t_pin, n * sizeof(int), 0) != hipSuccess ||
            hipHostMalloc((void**)&host, hostw * sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&host_dev, host, 0) != hipSuccess)
            return NOPE_ERR_ALLOC;
        memset(host, 0, hostw * sizeof(unsigned));
        t.assign(n, 0);
        moved.assign(n, 0);
        return NOPE_OK;
    }
    void destroy() {
        if (tab_pin) hipHostFree(tab_pin);
        if (t_pin) hipHostFree(t_pin);
        if (host) hipHostFree(host);
        tab_pin = t_pin = nullptr; host = nullptr;
    }

    // The verdicts that have reached the host since the previous poll (no synchronisation): NOPE_OK, NOPE_ERR_RANGE (a judged forward
    // was out of range: its output is NaN), NOPE_ERR_RANGE_F16 (an activation was infinite).  Re-centred shifts are enqueued on `s`.
    int poll(hipStream_t s, int* n_out_of_range, int* n_adjusted, float* max_abs) {
        if (n_out_of_range) *n_out_of_range = 0;
        if (n_adjusted) *n_adjusted = 0;
        if (max_abs) *max_abs = 0.f;
        if (!active()) return NOPE_OK;
        std::lock_guard<std::mutex> lock(mu);
        volatile unsigned* h = host;
        const unsigned serial = h[0];
        if (serial == seen_serial) return NOPE_OK;            // no forward has finished since the last look
        __sync_synchronize();
        const unsigned bad_total = h[2], inf_total = h[3];
        const int bad = (int)(bad_total - seen_bad), fatal = (int)(inf_total - seen_inf);
        seen_serial = serial; seen_bad = bad_total; seen_inf = inf_total;
        const size_t n = tails.size();
        int nmoved = 0;
        float worst = 0.f;
        for (size_t i = 0; i < n; ++i) {
            const unsigned cur = h[4 + n + i];
            if (cur && cur < 0x7f800000u) { float c; memcpy(&c, &cur, 4); if (c > worst) worst = c; }
            const unsigned b = h[4 + i];
            if (!b) continue;
            h[4 + i] = 0u;                                    // (taken; the device writes it again when the layer is flagged again)
            if (b >= 0x7f800000u) continue;                   // non-finite: nothing a shift repairs (the f32 path overflows there too)
            float a;
            memcpy(&a, &b, 4);
            if (a > worst) worst = a;
            int e = 0;
            frexpf(a, &e);                                    // a = m * 2^e, m in [0.5, 1)
            int tn = (e - 1) - 8;                             // max |a| * 2^-tn in [256, 512)
            tn = tn < -100 ? -100 : (tn > 100 ? 100 : tn);
            if (tn != t[i]) {
                t_pin[i] = tn;
                if (hipMemcpyAsync(tails[i] + 3, &t_pin[i], sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) return NOPE_ERR_LAUNCH;
                t[i] = tn;
                moved[i] = 1;
                ++nmoved;
            }
        }
        if (n_out_of_range) *n_out_of_range = bad;
        if (n_adjusted) *n_adjusted = nmoved;
        if (max_abs) *max_abs = worst;
        return fatal > 0 ? NOPE_ERR_RANGE_F16 : (bad > 0 ? NOPE_ERR_RANGE : NOPE_OK);
    }
    int shifts(int* out, int max, int* n) {
        std::lock_guard<std::mutex> lock(mu);
        *n = (int)t.size();
        for (int i = 0; i < *n && i < max; ++i) out[i] = t[i];
        return NOPE_OK;
    }
};

// One forward's bookkeeping (host side; every launch it makes goes to `s`)
struct X2Fwd {
    X2Range* r = nullptr;
    hipStream_t s = nullptr;
    bool on = false;                             // r->active() and this is not a workspace-size query
    int err = NOPE_OK;
    std::map<const void*, int> slot_of;          // which slot holds max |.| of the tensor that currently lives at a buffer address
    int next_slot = 0;
    std::vector<int> tab;                        // [layer][5] = {t, slots of the tensors the layer's two-pass launches of THIS forward read}

    unsigned* slot_ptr(int sl) const { return r->amax + (size_t)sl * kX2SlotWords; }
    int produce(const void* p) {                 // a kernel that records its output's maximum is about to write the tensor at p
        const int sl = next_slot < X2Range::SLOTS ? next_slot++ : -1;
        if (sl >= 0) slot_of[p] = sl; else slot_of.erase(p);
        return sl;
    }
    void overwritten(const void* p) { slot_of.erase(p); }      // ... a kernel that does not
    // the slot of the f32 tensor at p, taking an absmax pass over it when its producer recorded none
    int slot_for(const void* p, size_t elems) {
        auto it = slot_of.find(p);
        if (it != slot_of.end()) return it->second;
        const int sl = produce(p);
        if (sl >= 0) { const int e = launch_absmax_f32((const float*)p, elems, slot_ptr(sl), s); if (e && !err) err = e; }
        return sl;
    }
    void consumes(int layer, int sl) {           // a two-pass launch of `layer` reads the tensor of slot sl
        if (sl < 0) return;
        if (tab.empty()) tab.assign(r->tails.size() * 5, -1);
        int* row = &tab[(size_t)layer * 5];
        int k = 1;
        while (k < 5 && row[k] >= 0 && row[k] != sl) ++k;
        if (k < 5) row[k] = sl;                  // (a layer launched with more than four different inputs in one forward: none in these networks)
    }
    // behind the forward's last kernel: the verdict and, if a layer left its window, NaNs over the forward's output
    int finish(void* out, size_t out_bytes, int out_dt) {
        if (!on) return err;
        const size_t nl = r->tails.size();
        if (tab.empty()) tab.assign(nl * 5, -1);
        int* pin;
        {
            std::lock_guard<std::mutex> lock(r->mu);
            for (size_t l = 0; l < nl; ++l) tab[l * 5] = r->t[l];
            pin = r->tab_pin + (size_t)(r->ring_i++ % X2Range::RING) * nl * 5;
        }
        memcpy(pin, tab.data(), nl * 5 * sizeof(int));
        if (hipMemcpyAsync(r->tab_dev, pin, nl * 5 * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) return NOPE_ERR_LAUNCH;
        // (only the slots this forward handed out: the others are zero since the last verdict)
        int e = launch_x2_verdict(r->amax, next_slot > 0 ? next_slot : 1, r->tab_dev, (int)nl, r->status_dev, r->host_dev, s);
        if (!e && out && out_bytes % 4 == 0) e = launch_x2_poison(out, out_bytes, out_dt, r->status_dev, s);
        return e ? e : err;
    }
};

}  // namespace nope
