// Host-only helper of the read-ahead replay (sydr_amd/channel/readahead.py): which tick of the receiver's per-millisecond loop
// (receiver.py:120-131) releases which epoch of a block that was tracked ahead in one launch, and what every tick's
// CHANNEL_UPDATE packets report (channel.py:205-228) -- the arithmetic of channel.py:137-146 (an epoch runs in the first tick
// whose slab completes it) and channelManager.py:149-188 (one epoch per channel and tick), for a whole block at once.
// No device code: plain loops over [channels][epochs]; 32 x 52 takes microseconds where the same in NumPy array operations
// took 0.4 ms per block.
#include <algorithm>
#include <cstring>
#include <vector>

#include "engine_internal.h"

extern "C" int sdr_block_schedule(const sdr_track_epoch* records, int n_ch, int n_cols, const int32_t* done, const int64_t* unread_now,
                                  int64_t samples_per_tick, const int64_t* flags0, const int64_t* code_since0, int max_ticks,
                                  int32_t* first, int32_t* n_ticks_out, int32_t* order_rows, int32_t* order_cols, int32_t* starts,
                                  sdr_track_epoch* records_sorted, sdr_track_epoch* last_records, int64_t* unread, int64_t* dev_flags,
                                  int64_t* code_count, int32_t* last_tick, int32_t* bit_rows, int32_t* bit_cols, int32_t* bit_values,
                                  int32_t* n_bits_out) {
    if (!records || !done || !unread_now || !first || !n_ticks_out || !order_rows || !order_cols || !starts || !records_sorted ||
        !last_records || !unread || !dev_flags || !code_count || !last_tick || !flags0 || !code_since0 || !bit_rows || !bit_cols ||
        !bit_values || !n_bits_out)
        return sdr_fail(SDR_ERR_INVALID, "block schedule: NULL argument");
    if (n_ch < 1 || n_cols < 1 || samples_per_tick < 1 || max_ticks < 1) return sdr_fail(SDR_ERR_INVALID, "block schedule: empty block");
    const int64_t spt = samples_per_tick;
    int n_ticks = 0;
    // ---- the tick of every epoch: the first whose slab completes it, and at least one tick after the channel's previous epoch
    for (int r = 0; r < n_ch; ++r) {
        if (done[r] < 0 || done[r] > n_cols) return sdr_fail(SDR_ERR_RANGE, "block schedule: channel %d reports %d epochs of %d", r, done[r], n_cols);
        int64_t end = 0;
        int64_t prev = -1;
        last_tick[r] = -1;
        for (int e = 0; e < n_cols; ++e) {
            int32_t& f = first[(size_t)r * n_cols + e];
            if (e >= done[r]) {
                f = -1;
                continue;
            }
            end += records[(size_t)r * n_cols + e].n_samples;
            const int64_t need = end - unread_now[r];                     // samples still to arrive
            int64_t k = need <= 0 ? 0 : (need + spt - 1) / spt - 1;       // ceil(need / spt) - 1
            if (k < 0) k = 0;
            if (k < prev + 1) k = prev + 1;
            if (k >= max_ticks) return sdr_fail(SDR_ERR_RANGE, "block schedule: an epoch falls into tick %lld of at most %d", (long long)k, max_ticks);
            f = (int32_t)k;
            prev = k;
            last_tick[r] = (int32_t)k;
            if ((int)k + 1 > n_ticks) n_ticks = (int)k + 1;
        }
        last_records[r] = done[r] > 0 ? records[(size_t)r * n_cols + done[r] - 1] : sdr_track_epoch{};
    }
    *n_ticks_out = n_ticks;
    // ---- the navigation bits the block decided, channel by channel in epoch order
    int nb = 0;
    for (int r = 0; r < n_ch; ++r)
        for (int e = 0; e < done[r]; ++e) {
            const int bit = records[(size_t)r * n_cols + e].nav_bit;
            if (bit >= 0) bit_rows[nb] = r, bit_cols[nb] = e, bit_values[nb] = bit, ++nb;
        }
    *n_bits_out = nb;
    // ---- the epochs in the order of their ticks, channels ascending inside a tick (counting sort by tick)
    for (int k = 0; k <= n_ticks; ++k) starts[k] = 0;
    for (int r = 0; r < n_ch; ++r)
        for (int e = 0; e < done[r]; ++e) ++starts[first[(size_t)r * n_cols + e] + 1];
    for (int k = 0; k < n_ticks; ++k) starts[k + 1] += starts[k];
    std::vector<int32_t> at(starts, starts + n_ticks + 1);
    for (int r = 0; r < n_ch; ++r)
        for (int e = 0; e < done[r]; ++e) {
            const int32_t p = at[first[(size_t)r * n_cols + e]]++;
            order_rows[p] = r, order_cols[p] = e;
            records_sorted[p] = records[(size_t)r * n_cols + e];
        }
    // ---- per tick and channel: unread samples, the device's flag bits and the epochs counted so far, as they stand after the tick
    for (int r = 0; r < n_ch; ++r) {
        int64_t consumed = 0, count = 0, flags = flags0[r];
        int e = 0;
        for (int k = 0; k < n_ticks; ++k) {
            if (e < done[r] && first[(size_t)r * n_cols + e] == k) {
                const sdr_track_epoch& rec = records[(size_t)r * n_cols + e];
                consumed += rec.n_samples;
                flags = rec.track_flags;
                ++count;
                ++e;
            }
            unread[(size_t)k * n_ch + r] = unread_now[r] + (int64_t)(k + 1) * spt - consumed;
            dev_flags[(size_t)k * n_ch + r] = flags;
            code_count[(size_t)k * n_ch + r] = code_since0[r] + count;
        }
    }
    return SDR_OK;
}
