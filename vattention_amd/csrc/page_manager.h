// MI355X-native vAttention page manager — bookkeeping core + mapper thread.
// No torch, no HIP in this header: the physical backend is a vattn_backend_ops table.
//
// Bookkeeping follows the reference state machine line by line where it is observable
// (/root/reference/vattention/vattention.cu:189-609, utils.h:83-228, mux.h:1-85); what is new is
// the execution model: every bookkeeping routine only *plans* physical operations (PhysOp); the
// plan is then executed either before the API call returns (what the current step needs) or by a
// single mapper thread (look-ahead / reclamation), see include/vattn.h.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/vattn.h"

namespace vattn {

struct PhysOp {
    uint8_t kind;        // 0 = map page at (tensor, offset), 1 = unmap (tensor, offset)
    uint8_t need_fence;  // unmap of a FREED slot's page: kernels launched before the free may still read it (fence / quiesce first)
    uint16_t slot;       // request slot the page belongs to
    uint16_t layer;      // layer of the page (0 with megacache)
    uint32_t tensor;
    uint32_t page;       // physical page id (index into handles_)
    uint64_t offset;     // byte offset inside the tensor's virtual range
};

class PageManager {
public:
    PageManager(const vattn_config& cfg, const vattn_backend_ops& be);
    ~PageManager();

    int init();                                   // reserve virtual ranges, start the mapper
    int num_tensors() const { return (int)bases_.size(); }
    uint64_t tensor_base(int i) const { return bases_[i]; }
    void layout(vattn_layout* out) const;

    int64_t reserve_physical_pages(uint64_t free_memory);
    int step(const uint64_t* lens, uint32_t n, bool eager_reclaim);
    int step_async(const uint64_t* lens, uint32_t n);
    int wait();
    int alloc_new_batch_idx(uint64_t seqlen);
    int free_batch_idx(int slot, void* stream = nullptr, bool with_fence = false);
    int premap(uint64_t seqlen);
    int cancel_premap(int slot);
    int64_t wait_pool_ready(int64_t timeout_ms);      // include/vattn.h vattn_wait_pool_ready
    int wait_layer(uint32_t layer);
    uint32_t layers_ready();
    int set_sync_layers(uint32_t n);
    uint64_t num_free_kvblocks();
    int set_deferred_reclamation(bool on);
    int set_verbose(bool on);
    int map_common_pages(uint64_t num_tokens);
    int show_kvcache_config();
    int show_allocator_state();
    int cleanup();

    int64_t state_dump(uint64_t* out, uint64_t cap);
    int64_t pagemap_dump(uint64_t* out, uint64_t cap_rows);
    void stats(vattn_stats* out);
    void counts(uint64_t out[4]);
    const char* last_error() const { return last_error_.c_str(); }
    void set_error(const std::string& msg) { last_error_ = msg; }

private:
    // ---- configuration (vattention.cu:38-74) ----
    vattn_config cfg_;
    vattn_backend_ops be_;
    uint64_t tokens_per_page_ = 0, virt_per_req_ = 0, max_pages_per_req_ = 0, virt_total_ = 0;
    uint64_t row_bytes_ = 0;
    bool cleaned_ = false, inited_ = false;

    // ---- bookkeeping state (utils.h:12-81) ----
    std::vector<uint64_t> mapped_pages_, lens_;
    std::vector<uint32_t> pool_;                                  // LIFO of page ids
    std::map<std::tuple<uint64_t, uint64_t, uint64_t>, std::pair<uint32_t, uint32_t>> pagemap_;
    bool deferred_reclaim_ = true, verbose_ = false;
    uint64_t num_pages_ = 0;                                      // page ids handed out so far
    // Prefix sharing (map_common_pages aliases one physical pair into several slots): a page returns to the pool when its
    // LAST mapping goes (the reference pushes it once per slot, mux.h:51-66 — duplicate ids in the pool, the same physical
    // page then backs two different (slot, layer) ranges).  refcnt_[page] = live mappings; shared_ lists the aliased
    // page-groups and who still holds them, so that num_free_kvblocks counts a shared group once, and only when every
    // holder could release it.
    std::vector<uint32_t> refcnt_;
    struct SharedGroup { std::vector<std::pair<uint32_t, uint64_t>> holders; };   // (slot, page position inside the slot)
    std::vector<SharedGroup> shared_;
    std::vector<uint8_t> reserved_;                               // slots handed out by premap() and not yet activated / freed
    // inherited_[r]: the first inherited_[r] page positions of slot r were mapped when the slot was last FREED — kernels of the
    // previous occupant launched before that free may still read them (the engine frees right after launching its last iteration),
    // also after the slot has been handed to a new request.  Unmapping one of them needs the slot's fence (or a quiesce) whatever
    // the slot's current state; positions at or above it were mapped for the current occupant and follow the active-slot rule.
    std::vector<uint64_t> inherited_;
    std::atomic<int> fatal_{0};                                   // sticky: a failed unmap / set-access / TLB invalidation
    std::mutex state_mu_;
    std::vector<PhysOp> plan_;
    std::string last_error_;

    // ---- helpers mirrored from utils.h ----
    uint64_t tokens_to_pages(uint64_t n) const { return (n + tokens_per_page_ - 1) / tokens_per_page_; }
    bool active(int r) const { return lens_[r] != 0; }
    uint64_t pages_to_kvblocks(uint64_t pages) const { return cfg_.megacache ? pages / 2 : pages / (2ull * cfg_.num_layers); }
    bool kvblocks_available(uint64_t n) const { return pages_to_kvblocks(pool_.size()) >= n; }
    uint64_t need_new_page_async(int r, uint64_t eager) const;
    uint32_t k_tensor(uint32_t layer) const { return cfg_.megacache ? 0 : layer; }
    uint32_t v_tensor(uint32_t layer) const { return cfg_.megacache ? 1 : cfg_.num_layers + layer; }

    // ---- planners (each mirrors one reference routine) ----
    int plan_map_pair(int r, uint32_t layer, uint64_t off);
    void plan_unmap_pair(int r, uint32_t layer, uint64_t off, bool inherited);
    void drop_mapping(uint32_t page);          // refcount--, back to the pool at zero
    void forget_shared_holder(int r, uint64_t pos);
    void rollback_maps(const std::vector<PhysOp>& ops, size_t first_failed);   // state_mu_ held
    void unmap_req_page_one(int r);
    void release_some(int r, uint64_t retain);
    int grow(int r, uint64_t nblocks, bool sync);
    void reclaim_on_demand(uint64_t nblocks, bool allow_reserved = true);
    void do_reclaim_pages();
    int map_pages_for_curr_step(int r, uint64_t seq_len);
    void background_management();
    void log(const std::string& s) const;
    int fail(int code, const std::string& msg);

    // ---- executor ----
    std::vector<uint64_t> bases_;
    std::vector<uint64_t> handles_;              // page id -> backend handle (0 = not created yet)
    std::vector<uint8_t> created_;
    std::vector<uint32_t> create_order_;         // page ids in creation order: handles are released oldest-first (see cleanup)
    std::atomic<int> precreate_error_{0};       // VATTN_ERR_* of a creation ahead of demand that failed (wait_pool_ready reports it)
    std::atomic<uint64_t> precreate_left_{0};   // ids [0, precreate_left_) still to be looked at, top down
    std::atomic<uint64_t> join_wait_ns_{0};
    std::atomic<uint64_t> frontier_{0};          // lowest page id ever handed out by the pool (ids below it were never used)
    uint64_t precreate_floor() const {           // ids below this are not created ahead of demand (yet): sliding window
        const uint64_t f = frontier_.load(std::memory_order_relaxed);
        const uint64_t w = precreate_window_.load(std::memory_order_relaxed);
        return f > w ? f - w : 0;
    }
    // pages created ahead of demand by the idle mapper: the whole pool when it has few enough handles that creating them all
    // costs seconds (hipMemCreate is O(live handles): 31 k handles of 8 MiB = 6 s in the background), a sliding window otherwise
    static constexpr uint64_t kPrecreateAheadPages = 4096;
    static constexpr uint64_t kPrecreateWholePoolBelow = 40000;
    std::atomic<uint64_t> precreate_window_{kPrecreateAheadPages};
    std::mutex exec_mu_;                         // serialises driver calls
    std::mutex q_mu_;
    std::condition_variable q_cv_, done_cv_;
    std::deque<std::vector<PhysOp>> queue_;
    uint64_t inflight_ = 0;
    bool stop_ = false;
    std::thread mapper_;
    int async_error_ = 0;
    std::string async_error_msg_;
    vattn_stats st_{};
    // a background batch whose map failed: the joiner rolls its unexecuted maps back (bookkeeping is guarded by state_mu_,
    // which the mapper never takes)
    std::vector<std::vector<PhysOp>> failed_later_;   // batches queued behind a failed one: their maps were not tried (page_manager.cpp, mapper_main)
    std::vector<PhysOp> failed_ops_;
    size_t failed_at_ = 0;
    bool have_failed_ = false;
    bool failed_layered_ = false;                // the failed batch was the mapper's half of a layer-ordered step
    std::vector<PhysOp> layered_now_;            // the synchronous half (layers [0, sync_layers)) of the layer-ordered batch in flight:
                                                 // a failure of the mapper's half takes the WHOLE page-groups back, these included
    std::atomic<int> fg_waiting_{0};             // a foreground flush wants exec_mu_: the idle pre-creation loop backs off
    // VATTN_FLAG_LAYERED_ASYNC
    uint32_t sync_layers_ = 2;
    std::atomic<uint32_t> layers_ready_{0};      // layers of the current step whose pages are mapped
    std::atomic<int> layered_pending_{0};
    std::atomic<int> layered_error_{0};
    std::atomic<uint64_t> layer_wait_ns_{0};
    std::mutex layer_mu_;
    std::condition_variable layer_cv_;
    std::deque<uint8_t> queue_layered_;          // parallel to queue_: 1 = ops are sorted by layer, publish progress

    int flush_sync();
    void flush_async();
    int execute(const std::vector<PhysOp>& ops, bool is_async, size_t* first_failed, bool layered = false, bool skip_maps = false);
    void note_popped(uint32_t lowest);
    int ensure_created(uint32_t page);
    void mapper_main();
    int wait_locked_free();
};

}  // namespace vattn
