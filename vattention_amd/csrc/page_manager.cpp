// See page_manager.h.  Reference citations are relative to /root/reference/vattention/.
#include "page_manager.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <sstream>

namespace vattn {

namespace {
constexpr uint64_t kEagerNumSteps = 10;    // vattention.cu:484
constexpr uint64_t kEagerNumKvBlocks = 2;  // vattention.cu:485
// hipMemCreate is O(live handles) on ROCm 7.2 (9 us at 5 k handles, 106 us at 20 k, 382 us at 50 k, 931 us at 100 k:
// profiles/r01_vmm_scale_probe.txt), so materialising a whole pool up front is quadratic.  The idle mapper thread keeps
// PageManager::kPrecreateAheadPages handles created BELOW the lowest page id the pool has handed out so far (a window that
// slides down as the pool is consumed); a map that outruns the window creates its handle itself.

inline uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}
inline uint64_t round_up(uint64_t x, uint64_t y) { return ((x + y - 1) / y) * y; }  // utils.h:10
}  // namespace

PageManager::PageManager(const vattn_config& cfg, const vattn_backend_ops& be) : cfg_(cfg), be_(be) {}

PageManager::~PageManager() {
    if (inited_ && !cleaned_) cleanup();
    {
        std::lock_guard<std::mutex> l(q_mu_);
        stop_ = true;
    }
    q_cv_.notify_all();
    if (mapper_.joinable()) mapper_.join();
}

int PageManager::fail(int code, const std::string& msg) {
    last_error_ = msg;
    return code;
}

void PageManager::log(const std::string& s) const {
    if (verbose_) std::cout << s << std::endl;   // utils.h:230-238
}

// vattention.cu:97-128 (asserts :107-110 are explicit errors here), :38-74, utils.h:88-97
int PageManager::init() {
    if (!(cfg_.max_batch_size > 0 && cfg_.max_batch_size < 1000)) return fail(VATTN_ERR_INVALID, "max_batch_size must be in (0, 1000)");
    if (!(cfg_.max_context_length > 0 && cfg_.max_context_length < 1000000)) return fail(VATTN_ERR_INVALID, "max_context_length must be in (0, 1000000)");
    if (!(cfg_.num_layers > 0 && cfg_.num_layers < 100)) return fail(VATTN_ERR_INVALID, "num_layers must be in (0, 100)");
    if (!(cfg_.num_kv_heads > 0 && cfg_.num_kv_heads < 256)) return fail(VATTN_ERR_INVALID, "num_kv_heads must be in (0, 256)");
    if (cfg_.head_size == 0 || cfg_.itemsize == 0 || cfg_.page_size == 0) return fail(VATTN_ERR_INVALID, "head_size, itemsize and page_size must be non-zero");

    uint64_t min_gran = 0, rec_gran = 0;
    if (be_.granularity(be_.ctx, &min_gran, &rec_gran) != 0) return fail(VATTN_ERR_DRIVER, "granularity query failed");
    if (min_gran == 0 || cfg_.page_size % min_gran != 0) {
        std::ostringstream ss;
        ss << "page_size " << cfg_.page_size << " is not a multiple of the device's minimum mapping granularity " << min_gran;
        return fail(VATTN_ERR_INVALID, ss.str());
    }
    row_bytes_ = (uint64_t)cfg_.num_kv_heads * cfg_.head_size * cfg_.itemsize;
    if (cfg_.megacache) row_bytes_ *= cfg_.num_layers;
    tokens_per_page_ = cfg_.page_size / row_bytes_;
    if (tokens_per_page_ == 0) return fail(VATTN_ERR_INVALID, "page_size is smaller than one token's KV row");
    virt_per_req_ = round_up(row_bytes_ * cfg_.max_context_length, cfg_.page_size);
    max_pages_per_req_ = virt_per_req_ / cfg_.page_size;
    virt_total_ = virt_per_req_ * cfg_.max_batch_size;

    mapped_pages_.assign(cfg_.max_batch_size, 0);
    lens_.assign(cfg_.max_batch_size, 0);
    reserved_.assign(cfg_.max_batch_size, 0);
    inherited_.assign(cfg_.max_batch_size, 0);

    const int nt = cfg_.megacache ? 2 : 2 * (int)cfg_.num_layers;
    const uint64_t align = std::max<uint64_t>(cfg_.page_size, rec_gran);
    for (int i = 0; i < nt; i++) {
        uint64_t base = 0;
        if (be_.reserve_va(be_.ctx, virt_total_, align, &base) != 0 || base == 0) {
            for (uint64_t b : bases_) be_.free_va(be_.ctx, b, virt_total_);
            bases_.clear();
            std::ostringstream ss;
            ss << "virtual address reservation of " << virt_total_ << " bytes failed (tensor " << i << ")";
            return fail(VATTN_ERR_DRIVER, ss.str());
        }
        bases_.push_back(base);
    }
    if (!(cfg_.flags & VATTN_FLAG_NO_MAPPER_THREAD)) mapper_ = std::thread([this] { mapper_main(); });
    inited_ = true;
    return VATTN_OK;
}

void PageManager::layout(vattn_layout* o) const {
    const uint64_t kvh = cfg_.num_kv_heads, D = cfg_.head_size, L = cfg_.num_layers;
    const uint64_t bstride = virt_per_req_ / cfg_.itemsize;
    if (cfg_.megacache) {   // vattention.cu:146-147
        o->ndim = 5;
        const uint64_t sh[5] = {cfg_.max_batch_size, cfg_.max_context_length, L, kvh, D};
        const uint64_t st[5] = {bstride, L * kvh * D, kvh * D, D, 1};
        for (int i = 0; i < 5; i++) { o->shape[i] = sh[i]; o->stride[i] = st[i]; }
    } else {                // vattention.cu:149
        o->ndim = 4;
        const uint64_t sh[5] = {cfg_.max_batch_size, cfg_.max_context_length, kvh, D, 0};
        const uint64_t st[5] = {bstride, kvh * D, D, 1, 0};
        for (int i = 0; i < 5; i++) { o->shape[i] = sh[i]; o->stride[i] = st[i]; }
    }
    o->virt_bytes_per_req = virt_per_req_;
    o->virt_bytes_total = virt_total_;
    o->tokens_per_page = tokens_per_page_;
    o->max_pages_per_req = max_pages_per_req_;
    o->page_size = cfg_.page_size;
}

// ------------------------------------------------------------------------------------------
// bookkeeping planners
// ------------------------------------------------------------------------------------------

uint64_t PageManager::need_new_page_async(int r, uint64_t eager) const {   // utils.h:206-219
    if (!active(r)) return 0;
    const uint64_t mapped = mapped_pages_[r];
    if (mapped == max_pages_per_req_) return 0;
    const uint64_t req = tokens_to_pages(lens_[r] + eager);
    return req <= mapped ? 0 : req - mapped;
}

void PageManager::note_popped(uint32_t lowest) {     // state_mu_ held
    if (lowest < frontier_.load(std::memory_order_relaxed)) {
        {
            std::lock_guard<std::mutex> q(q_mu_);   // the mapper evaluates its wake-up predicate under q_mu_: no lost wake-up
            frontier_.store(lowest, std::memory_order_relaxed);
        }
        q_cv_.notify_one();        // the window slid down: the idle mapper may create more handles
    }
}

void PageManager::drop_mapping(uint32_t page) {      // state_mu_ held
    if (refcnt_[page] > 0) refcnt_[page]--;
    if (refcnt_[page] == 0) pool_.push_back(page);
}

void PageManager::forget_shared_holder(int r, uint64_t pos) {
    for (size_t i = 0; i < shared_.size(); i++) {
        auto& h = shared_[i].holders;
        for (size_t j = 0; j < h.size(); j++)
            if (h[j].first == (uint32_t)r && h[j].second == pos) {
                h.erase(h.begin() + j);
                if (h.empty()) shared_.erase(shared_.begin() + i);
                return;
            }
    }
}

int PageManager::plan_map_pair(int r, uint32_t layer, uint64_t off) {      // mux.h:37-48, cudaInternal.h:70-82
    if (pool_.size() < 2) return fail(VATTN_ERR_POOL_EMPTY, "***** page pool is empty *****");
    const uint32_t k = pool_.back(); pool_.pop_back();
    const uint32_t v = pool_.back(); pool_.pop_back();
    note_popped(k < v ? k : v);
    refcnt_[k] = refcnt_[v] = 1;
    plan_.push_back({0, 0, (uint16_t)r, (uint16_t)layer, k_tensor(layer), k, off});
    plan_.push_back({0, 0, (uint16_t)r, (uint16_t)layer, v_tensor(layer), v, off});
    pagemap_[std::make_tuple((uint64_t)r, off, (uint64_t)layer)] = std::make_pair(k, v);
    return VATTN_OK;
}

void PageManager::plan_unmap_pair(int r, uint32_t layer, uint64_t off, bool inherited) {   // mux.h:51-66
    // a FREED slot's pages may still be read by kernels launched before the free: fence (or quiesce) before the unmap — and so may
    // the pages a re-activated slot INHERITED from its previous occupant (`inherited`, see inherited_).  Pages mapped for an ACTIVE
    // slot's current occupant only go when its current length no longer needs them, and no kernel in flight reads beyond that.
    const uint8_t fence = (!active(r) || inherited) ? 1 : 0;
    auto key = std::make_tuple((uint64_t)r, off, (uint64_t)layer);
    auto it = pagemap_.find(key);
    const uint32_t k = it != pagemap_.end() ? it->second.first : 0, v = it != pagemap_.end() ? it->second.second : 0;
    plan_.push_back({1, fence, (uint16_t)r, (uint16_t)layer, k_tensor(layer), k, off});
    plan_.push_back({1, fence, (uint16_t)r, (uint16_t)layer, v_tensor(layer), v, off});
    if (it != pagemap_.end()) {
        drop_mapping(k);                        // K first, then V; a shared page only when its last mapping goes
        drop_mapping(v);
        pagemap_.erase(it);
    }
}

void PageManager::unmap_req_page_one(int r) {    // vattention.cu:219-241, utils.h:193-204
    const uint64_t off = (uint64_t)r * virt_per_req_ + (mapped_pages_[r] - 1) * cfg_.page_size;
    const bool inherited = mapped_pages_[r] - 1 < inherited_[r];
    if (cfg_.megacache) {
        plan_unmap_pair(r, 0, off, inherited);
    } else {
        for (uint32_t l = 0; l < cfg_.num_layers; l++) plan_unmap_pair(r, l, off, inherited);
    }
    mapped_pages_[r]--;
    if (inherited_[r] > mapped_pages_[r]) inherited_[r] = mapped_pages_[r];
    if (!shared_.empty()) forget_shared_holder(r, mapped_pages_[r]);
}

void PageManager::release_some(int r, uint64_t retain) {   // vattention.cu:243-247
    while (mapped_pages_[r] > retain) unmap_req_page_one(r);
}

int PageManager::grow(int r, uint64_t nblocks, bool sync) {   // vattention.cu:268-323
    if (nblocks == 0) return VATTN_OK;
    if (!kvblocks_available(nblocks)) {
        if (!sync) return VATTN_OK;               // background attempt: silently give up
        verbose_ = true;                          // the reference flips verbose on here (:283)
        log("free pages: " + std::to_string(pages_to_kvblocks(pool_.size())));
        log("required: " + std::to_string(nblocks));
        show_allocator_state();
        return fail(VATTN_ERR_OOM, "***** OOM on demand: not enough free pages to continue *****");
    }
    for (uint64_t c = 0; c < nblocks; c++) {
        const uint64_t off = (uint64_t)r * virt_per_req_ + mapped_pages_[r] * cfg_.page_size;   // utils.h:185-191
        if (!(off < (uint64_t)(r + 1) * virt_per_req_)) return VATTN_OK;   // is_valid_offset, :254-265 (never throws)
        if (cfg_.megacache) {
            int rc = plan_map_pair(r, 0, off);
            if (rc) return rc;
        } else {
            for (uint32_t l = 0; l < cfg_.num_layers; l++) {
                int rc = plan_map_pair(r, l, off);
                if (rc) return rc;
            }
        }
        mapped_pages_[r]++;
    }
    return VATTN_OK;
}

void PageManager::reclaim_on_demand(uint64_t nblocks, bool allow_reserved) {   // vattention.cu:420-438
    // (slots pre-mapped for the NEXT request — premap(), an extension — are only touched if nothing else can be reclaimed, and
    // never on behalf of another look-ahead)
    for (int pass = 0; pass < (allow_reserved ? 2 : 1); pass++)
        for (int r = (int)cfg_.max_batch_size - 1; r >= 0; r--) {
            if (kvblocks_available(nblocks)) return;
            if (pass == 0 && reserved_[r]) continue;
            if (pass == 1 && !reserved_[r]) continue;
            const uint64_t mapped = mapped_pages_[r];
            const uint64_t required = tokens_to_pages(lens_[r]);
            if (mapped <= required) continue;
            if (pass == 0) { release_some(r, required); continue; }
            while (mapped_pages_[r] > required && !kvblocks_available(nblocks)) unmap_req_page_one(r);   // only what is missing
        }
}

void PageManager::do_reclaim_pages() {   // vattention.cu:444-469
    if (deferred_reclaim_) return;
    int next_prefill = -1;
    for (int r = 0; r < (int)cfg_.max_batch_size; r++)
        if (!active(r)) { next_prefill = r; break; }
    for (int r = (int)cfg_.max_batch_size - 1; r >= 0; r--) {
        if (active(r) || r == next_prefill) continue;
        if (mapped_pages_[r] == 0) continue;
        unmap_req_page_one(r);
        break;
    }
}

int PageManager::map_pages_for_curr_step(int r, uint64_t seq_len) {   // vattention.cu:376-392
    uint64_t required = tokens_to_pages(seq_len);
    const uint64_t mapped = mapped_pages_[r];
    if (required <= mapped) return VATTN_OK;
    required -= mapped;
    if (!kvblocks_available(required)) reclaim_on_demand(required);
    log("[DEBUG] allocating " + std::to_string(required) + " pages for reqId: " + std::to_string(r));
    int rc = grow(r, required, true);
    if (rc) return rc;
    lens_[r] = seq_len;
    return VATTN_OK;
}

void PageManager::background_management() {   // vattention.cu:486-536
    uint64_t required = 0;
    for (int r = 0; r < (int)cfg_.max_batch_size; r++) required += need_new_page_async(r, 1);
    if (!kvblocks_available(required)) {
        log("[DEBUG] reclaiming " + std::to_string(required) + " KV blocks in background thread...");
        reclaim_on_demand(required);
    }
    if (!kvblocks_available(required)) return;
    uint64_t mapped_curr = 0;
    bool done = false;
    for (uint64_t eager = 1; eager < kEagerNumSteps && !done; eager++) {
        for (int r = 0; r < (int)cfg_.max_batch_size; r++) {
            const uint64_t n = need_new_page_async(r, eager);
            grow(r, n, false);
            mapped_curr += n;
            if (eager == 1) continue;
            if (mapped_curr >= kEagerNumKvBlocks) { done = true; break; }
        }
    }
    if (required) return;
    do_reclaim_pages();
}

// ------------------------------------------------------------------------------------------
// API
// ------------------------------------------------------------------------------------------

int64_t PageManager::reserve_physical_pages(uint64_t free_memory) {   // cudaInternal.h:45-59, utils.h:221-228
    std::lock_guard<std::mutex> l(state_mu_);
    if (!inited_ || cleaned_) return fail(VATTN_ERR_INVALID, "allocator is not initialised");
    uint64_t n = free_memory / cfg_.page_size;
    n -= n % (2ull * cfg_.num_layers);            // multiple of 2*L even in megacache mode
    if (n > 0xFFFFFFF0ull) return fail(VATTN_ERR_INVALID, "page count exceeds 32-bit page ids");
    if (fatal_.load()) return fail(fatal_.load(), last_error_);
    if (be_.mem_info && n > pool_.size()) {
        // The reference commits the memory here (cuMemCreate per page, cudaInternal.h:45-59).  Handles are created lazily in
        // this design, so refuse up front what the device could not back: a pool that only fails mid-serving is worse.
        uint64_t free_b = 0, total_b = 0;
        if (be_.mem_info(be_.ctx, &free_b, &total_b) == 0) {
            uint64_t uncreated = 0;
            for (uint32_t id : pool_) uncreated += created_[id] ? 0 : 1;
            const uint64_t want = (n - pool_.size() + uncreated) * cfg_.page_size;
            if (want > free_b) {
                std::ostringstream ss;
                ss << "reserve_physical_pages: " << want << " bytes of physical pages requested but only " << free_b
                   << " bytes of device memory are free (of " << total_b << ")";
                return fail(VATTN_ERR_OOM, ss.str());
            }
        }
    }
    if (n > 100000 && be_.tlb_flush) {            // a real (HIP) backend: creation cost grows with the number of live handles
        std::cerr << "[vattn] warning: " << n << " physical pages of " << (cfg_.page_size >> 10) << " KiB = one hipMemCreate handle each; "
                  << "handle creation on ROCm is O(live handles) (about 1 ms per call beyond 100 k, DESIGN.md section 3): prefer larger pages"
                  << (cfg_.megacache ? "" : " with the megacache layout") << " (8 MiB pages: " << (free_memory >> 23) << " handles)" << std::endl;
    }
    {
        std::lock_guard<std::mutex> e(exec_mu_);
        while (pool_.size() < n) {
            const uint32_t id = (uint32_t)num_pages_++;
            pool_.push_back(id);
            handles_.push_back(0);
            created_.push_back(0);
            refcnt_.push_back(0);
        }
    }
    if (cfg_.flags & VATTN_FLAG_EAGER_CREATE) {
        std::lock_guard<std::mutex> e(exec_mu_);
        for (uint32_t id = 0; id < num_pages_; id++) {
            int rc = ensure_created(id);
            if (rc) return fail(rc, async_error_msg_);
        }
    } else if (!(cfg_.flags & VATTN_FLAG_NO_MAPPER_THREAD)) {
        // lazy pool: the mapper materialises handles while it is idle, top of the LIFO first
        {
            std::lock_guard<std::mutex> q(q_mu_);
            frontier_.store(num_pages_);
            precreate_window_.store(num_pages_ <= kPrecreateWholePoolBelow ? num_pages_ : kPrecreateAheadPages);
            precreate_error_.store(0);
            precreate_left_.store(num_pages_);
        }
        q_cv_.notify_all();
    }
    return (int64_t)(int32_t)(uint32_t)pool_.size();   // apis.h:23 returns int
}

int PageManager::step(const uint64_t* lens, uint32_t n, bool eager_reclaim) {   // vattention.cu:395-409
    std::lock_guard<std::mutex> l(state_mu_);
    if (!inited_ || cleaned_) return fail(VATTN_ERR_INVALID, "allocator is not initialised");
    if (n != cfg_.max_batch_size) return fail(VATTN_ERR_INVALID, "seq_lens must have max_batch_size entries");
    if (fatal_.load()) return fail(fatal_.load(), last_error_);
    int rc = wait_locked_free();
    if (rc) return rc;
    int err = VATTN_OK;
    for (int r = 0; r < (int)cfg_.max_batch_size; r++) {
        lens_[r] = lens[r];
        if (lens[r] != 0) reserved_[r] = 0;
        if (eager_reclaim && lens[r] == 0 && mapped_pages_[r] != 0 && !reserved_[r]) {
            release_some(r, 0);
            continue;
        }
        err = map_pages_for_curr_step(r, lens[r]);
        if (err) break;
    }
    rc = flush_sync();
    return err ? err : rc;
}

int PageManager::step_async(const uint64_t* lens, uint32_t n) {   // vattention.cu:549-558
    std::lock_guard<std::mutex> l(state_mu_);
    if (!inited_ || cleaned_) return fail(VATTN_ERR_INVALID, "allocator is not initialised");
    if (n != cfg_.max_batch_size) return fail(VATTN_ERR_INVALID, "seq_lens must have max_batch_size entries");
    if (fatal_.load()) return fail(fatal_.load(), last_error_);
    lens_.assign(lens, lens + n);                       // utils.h:155-158
    for (uint32_t r = 0; r < n; r++)
        if (lens[r] != 0) reserved_[r] = 0;             // a pre-mapped slot has been claimed
    int rc = wait_locked_free();                        // wait_kvcache_manager_sync
    if (rc) return rc;
    int err = VATTN_OK;
    for (int r = 0; r < (int)cfg_.max_batch_size; r++) {   // prepare_prefill_kvcache, :411-418
        err = map_pages_for_curr_step(r, lens_[r]);
        if (err) break;
    }
    // VATTN_FLAG_LAYERED_ASYNC: what this step needs is split by layer — layers [0, sync_layers) are mapped before the
    // call returns, the rest by the mapper in layer order while those layers already run (vattn_wait_layer gates each
    // layer's kernels).  Only pure-map plans of a worthwhile size; plans that unmap stay synchronous.
    bool layered = false;
    if (!err && (cfg_.flags & VATTN_FLAG_LAYERED_ASYNC) && !(cfg_.flags & VATTN_FLAG_NO_MAPPER_THREAD) && !cfg_.megacache &&
        cfg_.num_layers > sync_layers_ && plan_.size() >= 4ull * cfg_.num_layers) {
        layered = true;
        for (const PhysOp& op : plan_)
            if (op.kind != 0) { layered = false; break; }
    }
    if (layered) {
        std::vector<PhysOp> now, later;
        for (const PhysOp& op : plan_) (op.layer < sync_layers_ ? now : later).push_back(op);
        std::stable_sort(later.begin(), later.end(), [](const PhysOp& a, const PhysOp& b) { return a.layer < b.layer; });
        plan_.clear();
        size_t failed_at = now.size();
        {
            fg_waiting_.fetch_add(1);
            std::lock_guard<std::mutex> e(exec_mu_);
            fg_waiting_.fetch_sub(1);
            rc = execute(now, false, &failed_at);
            if (!rc) st_.layered_batches++;
        }
        if (rc) {                                        // the synchronous layers failed: the whole step's plan is taken back
            last_error_ = async_error_msg_;
            if (failed_at < now.size()) {
                std::vector<PhysOp> all(now);
                all.insert(all.end(), later.begin(), later.end());
                rollback_maps(all, failed_at);
            }
            return rc;
        }
        layered_now_ = std::move(now);                   // mapper idle (joined above) and state_mu_ held: nobody reads it now
        layered_error_.store(0);
        layers_ready_.store(sync_layers_, std::memory_order_release);
        layered_pending_.store(1, std::memory_order_release);
        {
            std::lock_guard<std::mutex> q(q_mu_);
            queue_.push_back(std::move(later));
            queue_layered_.push_back(1);
            inflight_++;
        }
        q_cv_.notify_all();
    } else {
        rc = flush_sync();
        if (err) return err;
        if (rc) return rc;
    }
    background_management();                            // planned now, executed by the mapper
    flush_async();
    return VATTN_OK;
}

int PageManager::wait_layer(uint32_t layer) {
    if (!layered_pending_.load(std::memory_order_acquire)) return VATTN_OK;
    if (layers_ready_.load(std::memory_order_acquire) <= layer) {
        const uint64_t t0 = now_ns();
        std::unique_lock<std::mutex> l(layer_mu_);
        layer_cv_.wait(l, [&] { return layers_ready_.load(std::memory_order_acquire) > layer || !layered_pending_.load(); });
        layer_wait_ns_ += now_ns() - t0;
    }
    // (no message is written here: last_error_ belongs to the calls that hold state_mu_, and the engine thread may be inside
    // one of them; the C entry point returns only the code — vattn_last_error is NOT updated, include/vattn.h; the Python wrapper raises a fixed text — and the next step()/wait() reports the driver's message)
    return layered_error_.load();
}

uint32_t PageManager::layers_ready() {
    return layered_pending_.load(std::memory_order_acquire) ? layers_ready_.load(std::memory_order_acquire) : cfg_.num_layers;
}

int PageManager::set_sync_layers(uint32_t n) {
    std::lock_guard<std::mutex> l(state_mu_);
    if (n == 0) return fail(VATTN_ERR_INVALID, "sync_layers must be at least 1");
    sync_layers_ = n;
    return VATTN_OK;
}

int PageManager::wait() {
    std::lock_guard<std::mutex> l(state_mu_);
    return wait_locked_free();
}

int PageManager::alloc_new_batch_idx(uint64_t seqlen) {   // vattention.cu:564-589
    std::lock_guard<std::mutex> l(state_mu_);
    if (!inited_ || cleaned_) return -1;
    int new_id = -1;
    const uint64_t required = tokens_to_pages(seqlen);
    for (int r = 0; r < (int)cfg_.max_batch_size; r++) {
        if (active(r) || reserved_[r]) continue;
        if (new_id == -1) { new_id = r; continue; }
        if (mapped_pages_[r] >= required && mapped_pages_[r] < mapped_pages_[new_id]) new_id = r;
    }
    if (new_id != -1) lens_[new_id] = seqlen;
    return new_id;
}

// Admission look-ahead (MI355X extension; the reference's background thread only looks ahead on ACTIVE requests' growth,
// vattention.cu:486-536): picks the slot alloc_new_batch_idx(seqlen) would pick, reserves it for the caller and lets the MAPPER
// thread map the pages `seqlen` tokens need while the current iteration runs; the slot stays inactive (length 0) until the engine
// passes its length to step()/step_async(), which then finds the pages in place and maps nothing on the critical path.  The
// bookkeeping states it produces are ordinary ones (an inactive slot with mapped pages, as deferred reclamation leaves them).
int PageManager::premap(uint64_t seqlen) {
    std::lock_guard<std::mutex> l(state_mu_);
    if (!inited_ || cleaned_ || fatal_.load()) return -1;
    int slot = -1;
    const uint64_t required = tokens_to_pages(seqlen);
    for (int r = 0; r < (int)cfg_.max_batch_size; r++) {
        if (active(r) || reserved_[r]) continue;
        if (slot == -1) { slot = r; continue; }
        if (mapped_pages_[r] >= required && mapped_pages_[r] < mapped_pages_[slot]) slot = r;
    }
    if (slot < 0) return -1;
    reserved_[slot] = 1;
    // (the previous occupant's fence stays: it still guards the pages this slot inherits — inherited_ — until they are unmapped or
    // the next free records a new one; pages mapped from here on have no reader before the slot is activated)
    if (required > mapped_pages_[slot] && required <= max_pages_per_req_) {
        const uint64_t need = required - mapped_pages_[slot];
        // pool dry: take the pages from finished, unreserved slots HERE — the unmaps (and the wait for those slots' fences) then
        // run on the mapper thread under the current iteration, not in the activating step's critical path
        if (!kvblocks_available(need)) reclaim_on_demand(need, false);
        if (kvblocks_available(need)) {
            grow(slot, need, false);
            flush_async();
        } else if (!plan_.empty()) {
            flush_async();                              // partial reclaim: keep what was freed
        }
    }
    return slot;
}

int PageManager::cancel_premap(int slot) {
    std::lock_guard<std::mutex> l(state_mu_);
    if (slot < 0 || slot >= (int)cfg_.max_batch_size) return fail(VATTN_ERR_INVALID, "slot out of range");
    reserved_[slot] = 0;                                // the pages stay mapped: reclaimable like any finished slot's
    return VATTN_OK;
}

// include/vattn.h vattn_wait_pool_ready.  Reads atomics only (no manager lock: the engine thread may sit here while nothing else runs,
// and a test may call it beside step_async); the mapper's idle loop is what makes progress.
int64_t PageManager::wait_pool_ready(int64_t timeout_ms) {
    const uint64_t t0 = now_ns();
    for (;;) {
        const uint64_t left = precreate_left_.load(), floor = precreate_floor();
        if (precreate_error_.load() != 0) return (int64_t)precreate_error_.load();      // hipMemCreate failed ahead of demand (a failed creation also zeroes precreate_left_)
        if (fatal_.load()) return (int64_t)VATTN_ERR_DRIVER;
        if (left <= floor) return 0;
        if (timeout_ms >= 0 && now_ns() - t0 >= (uint64_t)timeout_ms * 1000000ull) return (int64_t)(left - floor);
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

int PageManager::free_batch_idx(int slot, void* stream, bool with_fence) {   // vattention.cu:591-594
    std::lock_guard<std::mutex> l(state_mu_);
    if (slot < 0 || slot >= (int)cfg_.max_batch_size) return fail(VATTN_ERR_INVALID, "slot out of range");
    lens_[slot] = 0;
    reserved_[slot] = 0;
    inherited_[slot] = mapped_pages_[slot];             // whoever gets the slot next inherits these, possibly still being read
    // the point in the engine's stream after the last kernel that can read this slot's pages (plain free: no fence, a later
    // reclaim of the slot then synchronises the whole device)
    if (be_.fence_record && be_.fence_record(be_.ctx, (uint32_t)slot, with_fence ? (stream ? stream : (void*)-1) : nullptr) != 0)
        return fail(VATTN_ERR_DRIVER, "recording the slot fence failed");
    return VATTN_OK;
}

uint64_t PageManager::num_free_kvblocks() {   // vattention.cu:189-210, utils.h:177-183 (u64 wrap kept)
    std::lock_guard<std::mutex> l(state_mu_);
    uint64_t over = 0;
    for (int r = 0; r < (int)cfg_.max_batch_size; r++) over += mapped_pages_[r] - tokens_to_pages(lens_[r]);
    // a shared page-group (map_common_pages) sits in several slots but is ONE group of physical pages: it is free-able once,
    // and only if every slot that holds it could release it
    for (const SharedGroup& g : shared_) {
        uint64_t recl = 0;
        for (auto& h : g.holders)
            if (h.second >= tokens_to_pages(lens_[h.first]) && h.second < mapped_pages_[h.first]) recl++;
        over -= recl;
        if (recl == g.holders.size() && recl) over += 1;
    }
    return pages_to_kvblocks(pool_.size()) + over;
}

int PageManager::set_deferred_reclamation(bool on) {
    std::lock_guard<std::mutex> l(state_mu_);
    deferred_reclaim_ = on;
    return VATTN_OK;
}

int PageManager::set_verbose(bool on) {
    std::lock_guard<std::mutex> l(state_mu_);
    verbose_ = on;
    return VATTN_OK;
}

int PageManager::map_common_pages(uint64_t num_tokens) {   // vattention.cu:325-373, mux.h:68-85
    std::lock_guard<std::mutex> l(state_mu_);
    if (!inited_ || cleaned_) return fail(VATTN_ERR_INVALID, "allocator is not initialised");
    if (fatal_.load()) return fail(fatal_.load(), last_error_);
    int rc = wait_locked_free();
    if (rc) return rc;
    const uint64_t nblocks = tokens_to_pages(num_tokens);
    if (nblocks == 0) return VATTN_OK;
    if (!kvblocks_available(nblocks)) return fail(VATTN_ERR_OOM, "***** OOM on demand: not enough free pages to continue *****");
    // The reference has no bound here (no is_valid_offset in map_common_pages_in_batch, vattention.cu:325-373): a slot that
    // already holds max_pages_per_req pages would be mapped INTO THE NEXT SLOT's range (or past the reservation).  Refuse.
    for (uint32_t r = 0; r < cfg_.max_batch_size; r++)
        if (mapped_pages_[r] + nblocks > max_pages_per_req_)
            return fail(VATTN_ERR_INVALID, "map_common_pages: a slot would exceed max_pages_per_req");
    int err = VATTN_OK;
    for (uint64_t c = 0; c < nblocks && !err; c++) {
        const uint32_t nl = cfg_.megacache ? 1 : cfg_.num_layers;
        for (uint32_t layer = 0; layer < nl && !err; layer++) {
            if (pool_.size() < 2) { err = fail(VATTN_ERR_POOL_EMPTY, "***** page pool is empty *****"); break; }
            const uint32_t k = pool_.back(); pool_.pop_back();
            const uint32_t v = pool_.back(); pool_.pop_back();
            note_popped(k < v ? k : v);
            refcnt_[k] = refcnt_[v] = cfg_.max_batch_size;          // one physical pair, max_batch_size mappings
            for (int r = 0; r < (int)cfg_.max_batch_size; r++) {
                const uint64_t off = (uint64_t)r * virt_per_req_ + mapped_pages_[r] * cfg_.page_size;
                plan_.push_back({0, 0, (uint16_t)r, (uint16_t)layer, k_tensor(layer), k, off});
                plan_.push_back({0, 0, (uint16_t)r, (uint16_t)layer, v_tensor(layer), v, off});
                pagemap_[std::make_tuple((uint64_t)r, off, (uint64_t)layer)] = std::make_pair(k, v);
            }
        }
        if (!err) {
            SharedGroup g;
            for (int r = 0; r < (int)cfg_.max_batch_size; r++) {
                g.holders.emplace_back((uint32_t)r, mapped_pages_[r]);
                mapped_pages_[r]++;
            }
            shared_.push_back(std::move(g));
        }
    }
    rc = flush_sync();
    return err ? err : rc;
}

int PageManager::show_kvcache_config() {   // vattention.cu:130-140
    std::lock_guard<std::mutex> l(state_mu_);
    log("Num layers: " + std::to_string(cfg_.num_layers));
    log("Num kv_heads: " + std::to_string(cfg_.num_kv_heads));
    log("Head size: " + std::to_string(cfg_.head_size));
    log("Max batch size: " + std::to_string(cfg_.max_batch_size));
    log("Max context length: " + std::to_string(cfg_.max_context_length));
    log("Bytes per elem: " + std::to_string(cfg_.itemsize));
    log("virt_buff_size_per_req: " + std::to_string(virt_per_req_));
    log("virt_buff_size: " + std::to_string(virt_total_));
    return VATTN_OK;
}

int PageManager::show_allocator_state() {   // vattention.cu:76-95 (caller holds state_mu_ or is internal)
    log("Free pool: " + std::to_string(pages_to_kvblocks(pool_.size())) + " KV blocks");
    log("reqId : seqlen: mapped: required");
    for (int i = 0; i < (int)cfg_.max_batch_size; i++) {
        std::ostringstream ss;
        ss << std::setw(8) << i << ": " << std::setw(8) << lens_[i] << " : " << std::setw(8) << mapped_pages_[i]
           << " : " << std::setw(8) << tokens_to_pages(lens_[i]);
        log(ss.str());
    }
    return VATTN_OK;
}

int PageManager::cleanup() {   // vattention.cu:601-609, mux.h:24-35, cudaInternal.h:84-94
    std::lock_guard<std::mutex> l(state_mu_);
    if (!inited_ || cleaned_) return VATTN_OK;
    wait_locked_free();
    for (int r = 0; r < (int)cfg_.max_batch_size; r++) release_some(r, 0);
    int rc = flush_sync();
    {
        std::lock_guard<std::mutex> e(exec_mu_);
        for (uint64_t b : bases_) be_.free_va(be_.ctx, b, virt_total_);
        bases_.clear();
        // oldest handle first: hipMemRelease costs O(position from the oldest live handle) on ROCm 7.2 [measured,
        // tools/vmm_release_probe.cpp: 40 k handles oldest-first 2.7 s, newest-first 5.9 s; page-id order IS newest-first here
        // because ids leave the LIFO pool from the top: 123 680 handles took 92 s]
        for (uint32_t i : create_order_) {
            if (created_[i]) {
                be_.release(be_.ctx, handles_[i]);
                created_[i] = 0;
                st_.handles_released++;
            }
        }
        create_order_.clear();
        precreate_left_.store(0);
    }
    pool_.clear();
    cleaned_ = true;
    log("released memory and cleaned up vattention ...");
    return rc;
}

int64_t PageManager::state_dump(uint64_t* out, uint64_t cap) {
    std::lock_guard<std::mutex> l(state_mu_);
    const uint64_t B = cfg_.max_batch_size;
    const uint64_t need = 3 + 2 * B + pool_.size();
    if (cap < need) return -(int64_t)need;
    uint64_t k = 0;
    out[k++] = B;
    out[k++] = pool_.size();
    out[k++] = pagemap_.size();
    for (uint64_t i = 0; i < B; i++) out[k++] = mapped_pages_[i];
    for (uint64_t i = 0; i < B; i++) out[k++] = lens_[i];
    for (uint32_t p : pool_) out[k++] = p;
    return (int64_t)k;
}

int64_t PageManager::pagemap_dump(uint64_t* out, uint64_t cap_rows) {
    std::lock_guard<std::mutex> l(state_mu_);
    uint64_t n = 0;
    for (auto& kv : pagemap_) {
        if (n >= cap_rows) return -(int64_t)pagemap_.size();
        out[5 * n + 0] = std::get<0>(kv.first);
        out[5 * n + 1] = std::get<1>(kv.first);
        out[5 * n + 2] = std::get<2>(kv.first);
        out[5 * n + 3] = kv.second.first;
        out[5 * n + 4] = kv.second.second;
        n++;
    }
    return (int64_t)n;
}

void PageManager::counts(uint64_t out[4]) {
    std::lock_guard<std::mutex> l(state_mu_);
    out[0] = pool_.size();
    out[1] = out[2] = out[3] = 0;
    for (uint32_t r = 0; r < cfg_.max_batch_size; r++) {
        out[1] += mapped_pages_[r];
        out[2] += tokens_to_pages(lens_[r]);
        out[3] += lens_[r] != 0;
    }
}

void PageManager::stats(vattn_stats* out) {
    std::lock_guard<std::mutex> e(exec_mu_);
    *out = st_;
    out->join_wait_ns = join_wait_ns_.load();
    out->layer_wait_ns = layer_wait_ns_.load();
}

// ------------------------------------------------------------------------------------------
// executor
// ------------------------------------------------------------------------------------------

int PageManager::ensure_created(uint32_t page) {   // exec_mu_ held
    if (created_[page]) return VATTN_OK;
    const uint64_t t0 = now_ns();
    uint64_t h = 0;
    if (be_.create(be_.ctx, cfg_.page_size, &h) != 0) {
        async_error_msg_ = "physical page allocation failed (out of device memory?)";
        return VATTN_ERR_DRIVER;
    }
    handles_[page] = h;
    created_[page] = 1;
    create_order_.push_back(page);
    st_.handles_created++;
    st_.create_ns += now_ns() - t0;
    return VATTN_OK;
}

int PageManager::execute(const std::vector<PhysOp>& ops, bool is_async, size_t* first_failed, bool layered, bool skip_maps) {   // exec_mu_ held
    const uint64_t t0 = now_ns();
    const uint64_t page = cfg_.page_size;
    const bool merge = !(cfg_.flags & VATTN_FLAG_NO_ACCESS_MERGE);
    if (first_failed) *first_failed = ops.size();
    // pending set-access runs: (tensor, start offset, bytes); flushed before any unmap, at every layer boundary of a
    // layer-ordered batch and at the end
    std::vector<std::tuple<uint32_t, uint64_t, uint64_t>> runs;
    auto flush_access = [&]() -> int {
        if (runs.empty()) return 0;
        if (merge) {
            std::sort(runs.begin(), runs.end());
            size_t w = 0;
            for (size_t i = 1; i < runs.size(); i++) {
                auto& a = runs[w];
                auto& b = runs[i];
                if (std::get<0>(a) == std::get<0>(b) && std::get<1>(a) + std::get<2>(a) == std::get<1>(b))
                    std::get<2>(a) += std::get<2>(b);
                else
                    runs[++w] = b;
            }
            runs.resize(w + 1);
        }
        for (auto& r : runs) {
            if (be_.set_access(be_.ctx, bases_[std::get<0>(r)] + std::get<1>(r), std::get<2>(r)) != 0) return -1;
            st_.access_calls++;
        }
        runs.clear();
        return 0;
    };
    auto publish_layer = [&](uint32_t ready) {
        {
            std::lock_guard<std::mutex> l(layer_mu_);
            layers_ready_.store(ready, std::memory_order_release);
        }
        layer_cv_.notify_all();
    };
    int rc = VATTN_OK;          // first error
    bool map_failed = skip_maps; // a create / map failed: the remaining maps are skipped (and rolled back by the caller),
                                // the remaining unmaps — whose bookkeeping is already applied — still run.  skip_maps: an EARLIER batch of
                                // this join window failed — this one's maps are not even tried (they sit above positions the joiner
                                // is about to take back), only its unmaps run
    if (skip_maps && first_failed) *first_failed = 0;
    bool unmapped = false, quiesced = false;
    const uint64_t c0_ns = st_.create_ns, c0_n = st_.handles_created, f0_ns = st_.fence_wait_ns + st_.quiesce_ns, m0 = st_.map_calls, u0 = st_.unmap_calls;
    uint64_t tlb_ns = 0;
    std::vector<uint8_t> fenced;    // slots whose fence this batch has already waited on
    for (size_t i = 0; i < ops.size(); i++) {
        const PhysOp& op = ops[i];
        if (op.kind == 0) {
            if (map_failed) continue;
            if (layered && i > 0 && ops[i - 1].layer != op.layer) {
                if (flush_access() != 0) { async_error_msg_ = "hipMemSetAccess failed"; rc = VATTN_ERR_DRIVER; fatal_ = rc; break; }
                publish_layer(op.layer);     // every layer below op.layer is mapped and accessible
            }
            int e = ensure_created(op.page);
            if (!e && be_.map(be_.ctx, bases_[op.tensor] + op.offset, page, handles_[op.page]) != 0) {
                async_error_msg_ = "hipMemMap failed";
                e = VATTN_ERR_DRIVER;
            }
            if (e) {
                rc = e;
                map_failed = true;
                if (first_failed) *first_failed = i;
                continue;
            }
            st_.map_calls++;
            st_.pages_mapped_now++;
            runs.emplace_back(op.tensor, op.offset, page);
        } else {
            if (flush_access() != 0) { async_error_msg_ = "hipMemSetAccess failed"; rc = VATTN_ERR_DRIVER; fatal_ = rc; break; }
            if (op.need_fence && !quiesced) {
                // kernels launched before the slot was freed may still read the page: wait for the slot's fence, or, if the
                // engine recorded none, for the whole device (once per batch)
                if (fenced.size() <= op.slot) fenced.resize((size_t)op.slot + 1, 0);
                if (!fenced[op.slot]) {
                    const uint64_t q0 = now_ns();
                    int fr = be_.fence_wait ? be_.fence_wait(be_.ctx, op.slot) : 1;
                    if (fr == 0) {
                        st_.fence_waits++;
                        st_.fence_wait_ns += now_ns() - q0;
                    } else if (fr > 0 && be_.quiesce) {
                        if (be_.quiesce(be_.ctx) != 0) fr = -1;
                        quiesced = true;
                        st_.quiesce_calls++;
                        st_.quiesce_ns += now_ns() - q0;
                    }
                    if (fr < 0) { async_error_msg_ = "device synchronisation before unmap failed"; rc = VATTN_ERR_DRIVER; fatal_ = rc; break; }
                    fenced[op.slot] = 1;
                }
            }
            if (be_.unmap(be_.ctx, bases_[op.tensor] + op.offset, page) != 0) {
                async_error_msg_ = "hipMemUnmap failed";
                rc = VATTN_ERR_DRIVER;
                fatal_ = rc;
                break;
            }
            st_.unmap_calls++;
            st_.pages_mapped_now--;
            unmapped = true;
        }
    }
    if (!fatal_ && flush_access() != 0) { async_error_msg_ = "hipMemSetAccess failed"; rc = VATTN_ERR_DRIVER; fatal_ = rc; }
    if (unmapped && be_.tlb_flush) {
        // stale GPU translations of the unmapped pages must be gone before anyone may rely on this batch
        const uint64_t f0 = now_ns();
        if (be_.tlb_flush(be_.ctx) != 0) {
            if (!rc) { async_error_msg_ = "TLB invalidation after unmap failed"; rc = VATTN_ERR_DRIVER; }
            fatal_ = VATTN_ERR_DRIVER;
        }
        st_.tlb_flushes++;
        tlb_ns = now_ns() - f0;
        st_.tlb_flush_ns += tlb_ns;
    }
    if (layered) {
        if (rc) layered_error_.store(rc);
        {
            std::lock_guard<std::mutex> l(layer_mu_);
            layers_ready_.store(cfg_.num_layers, std::memory_order_release);
            layered_pending_.store(0, std::memory_order_release);
        }
        layer_cv_.notify_all();
    }
    const uint64_t dt = now_ns() - t0;
    if (is_async) { st_.async_batches++; st_.async_ns += dt; } else {
        st_.sync_batches++;
        st_.sync_ns += dt;
        st_.sync_create_ns += st_.create_ns - c0_ns;
        st_.sync_creates += st_.handles_created - c0_n;
        st_.sync_fence_ns += st_.fence_wait_ns + st_.quiesce_ns - f0_ns;
        st_.sync_tlb_ns += tlb_ns;
        st_.sync_maps += st_.map_calls - m0;
        st_.sync_unmaps += st_.unmap_calls - u0;
    }
    return rc;
}

// A create / map failed at ops[first_failed]: every page-group that contains a map at or after that op is taken back —
// executed maps of those groups are unmapped, their pages return to the pool, the page map and the per-slot counts are
// restored — so bookkeeping equals the driver state again and the manager stays usable (the reference exits the process
// on any driver error, cudaInternal.h:1-13).  state_mu_ held; takes exec_mu_ for the unmaps.
void PageManager::rollback_maps(const std::vector<PhysOp>& ops, size_t first_failed) {
    std::vector<std::pair<uint16_t, uint64_t>> groups;      // (slot, page position) to revert
    auto pos_of = [&](const PhysOp& op) { return (op.offset - (uint64_t)op.slot * virt_per_req_) / cfg_.page_size; };
    for (size_t j = first_failed; j < ops.size(); j++) {
        if (ops[j].kind != 0) continue;
        const auto g = std::make_pair(ops[j].slot, pos_of(ops[j]));
        if (std::find(groups.begin(), groups.end(), g) == groups.end()) groups.push_back(g);
    }
    if (groups.empty()) return;
    std::lock_guard<std::mutex> e(exec_mu_);
    bool quiesced = false;
    for (size_t jj = ops.size(); jj-- > 0;) {
        const PhysOp& op = ops[jj];
        if (op.kind != 0) continue;
        const auto g = std::make_pair(op.slot, pos_of(op));
        if (std::find(groups.begin(), groups.end(), g) == groups.end()) continue;
        if (jj < first_failed) {                             // this one reached the driver: take it out again
            // ... but kernels of the running iteration may already read it (the synchronously mapped first layers of a layer-ordered
            // step): the device drains once before the first unmap of a rollback (an error path: the cost does not matter)
            if (!quiesced && be_.quiesce) {
                (void)be_.quiesce(be_.ctx);
                st_.quiesce_calls++;
                quiesced = true;
            }
            if (be_.unmap(be_.ctx, bases_[op.tensor] + op.offset, cfg_.page_size) == 0) {
                st_.unmap_calls++;
                st_.pages_mapped_now--;
            } else {
                fatal_ = VATTN_ERR_DRIVER;
            }
        }
        pagemap_.erase(std::make_tuple((uint64_t)op.slot, op.offset, (uint64_t)op.layer));
        drop_mapping(op.page);
    }
    for (auto& g : groups) {
        if (mapped_pages_[g.first] > g.second) mapped_pages_[g.first] = g.second;    // the reverted groups are a slot's tail
        if (!shared_.empty()) forget_shared_holder(g.first, g.second);
    }
    st_.rollbacks++;
}

int PageManager::wait_locked_free() {   // state_mu_ held; joins every queued background batch
    const uint64_t t0 = now_ns();
    std::vector<PhysOp> failed;
    std::vector<std::vector<PhysOp>> later;
    size_t failed_at = 0;
    int e = 0;
    {
        std::unique_lock<std::mutex> q(q_mu_);
        done_cv_.wait(q, [this] { return inflight_ == 0; });
        join_wait_ns_ += now_ns() - t0;
        later.swap(failed_later_);
        if (have_failed_) {
            failed.swap(failed_ops_);
            failed_at = failed_at_;
            have_failed_ = false;
            if (failed_layered_) {
                // the groups' first layers were mapped synchronously by step_async: they go back with the rest (otherwise they
                // stay mapped in the driver and in pagemap_ under a slot whose mapped_pages_ no longer covers them — the next
                // grow at that position would map over them and fail for good)
                failed.insert(failed.begin(), layered_now_.begin(), layered_now_.end());
                failed_at += layered_now_.size();
                failed_layered_ = false;
            }
        }
        e = async_error_;
        if (e) last_error_ = async_error_msg_;
        if (!fatal_) async_error_ = 0;          // reported once; bookkeeping is consistent again after the rollback below
    }
    for (size_t i = later.size(); i-- > 0;) rollback_maps(later[i], 0);      // youngest first: each batch maps above the one before it
    if (!failed.empty()) rollback_maps(failed, failed_at);
    return e;
}

int PageManager::flush_sync() {   // state_mu_ held, mapper idle (callers join first)
    if (plan_.empty()) return VATTN_OK;
    std::vector<PhysOp> ops;
    ops.swap(plan_);
    size_t failed_at = ops.size();
    int rc;
    {
        fg_waiting_.fetch_add(1);
        std::lock_guard<std::mutex> e(exec_mu_);
        fg_waiting_.fetch_sub(1);
        rc = execute(ops, false, &failed_at);
        if (rc) last_error_ = async_error_msg_;
    }
    if (rc && failed_at < ops.size()) rollback_maps(ops, failed_at);
    return rc;
}

void PageManager::flush_async() {   // state_mu_ held
    if (plan_.empty()) return;
    std::vector<PhysOp> ops;
    ops.swap(plan_);
    if (cfg_.flags & VATTN_FLAG_NO_MAPPER_THREAD) {
        size_t failed_at = ops.size();
        int rc;
        {
            std::lock_guard<std::mutex> e(exec_mu_);
            rc = execute(ops, true, &failed_at);
        }
        if (rc) {
            async_error_ = rc;
            if (failed_at < ops.size()) rollback_maps(ops, failed_at);
        }
        return;
    }
    {
        std::lock_guard<std::mutex> q(q_mu_);
        queue_.push_back(std::move(ops));
        queue_layered_.push_back(0);
        inflight_++;
    }
    q_cv_.notify_all();
}

void PageManager::mapper_main() {
    if (be_.thread_init) be_.thread_init(be_.ctx);
    std::unique_lock<std::mutex> q(q_mu_);
    for (;;) {
        q_cv_.wait(q, [this] { return stop_ || !queue_.empty() || precreate_left_.load() > precreate_floor(); });
        if (stop_) return;
        if (!queue_.empty()) {
            std::vector<PhysOp> ops = std::move(queue_.front());
            const bool layered = queue_layered_.front() != 0;
            queue_.pop_front();
            queue_layered_.pop_front();
            const bool after_failure = have_failed_;      // read under q_mu_ (the joiner resets it under q_mu_ once inflight_ == 0)
            q.unlock();
            int rc;
            size_t failed_at = ops.size();
            {
                std::lock_guard<std::mutex> e(exec_mu_);
                rc = execute(ops, true, &failed_at, layered, after_failure);
            }
            q.lock();
            if (rc && !async_error_) async_error_ = rc;
            if (after_failure) {
                // a batch queued BEHIND a failed one (step_async queues the look-ahead right after the layer-ordered half): its maps lie
                // at or above the positions the joiner takes back from the failed batch — executed, they would stay mapped in the
                // driver and in pagemap_ under a slot whose count no longer covers them.  None of them ran; the joiner reverts them all.
                bool any_map = false;
                for (const PhysOp& op : ops) any_map |= op.kind == 0;
                if (any_map) failed_later_.emplace_back(std::move(ops));
            } else if (rc && failed_at < ops.size()) {      // the joiner rolls the unexecuted maps back
                failed_ops_ = std::move(ops);
                failed_at_ = failed_at;
                have_failed_ = true;
                failed_layered_ = layered;
            }
            inflight_--;
            if (inflight_ == 0) done_cv_.notify_all();
            continue;
        }
        // idle: materialise physical handles ahead of demand (lazy pool).  ONE handle per acquisition of
        // exec_mu_ (a create costs ~1 ms at 100 k live handles) and none while a foreground flush is waiting for the mutex
        // (std::mutex is not fair), so step()/step_async() never queue behind a slice of creations.
        q.unlock();
        if (fg_waiting_.load() == 0) {
            std::lock_guard<std::mutex> e(exec_mu_);
            const uint64_t left = precreate_left_.load();
            if (left != 0 && left > precreate_floor()) {
                if (left > handles_.size() || ensure_created((uint32_t)(left - 1)) != 0) {
                    // creation ahead of demand failed (out of device memory: the reference aborts inside reserve, cudaInternal.h:45-59):
                    // remembered for wait_pool_ready; the pages are created on demand from here on and a step that needs one reports it
                    precreate_error_.store(VATTN_ERR_DRIVER);
                    precreate_left_.store(0);
                } else precreate_left_.store(left - 1);
            }
        } else {
            std::this_thread::yield();
        }
        q.lock();
    }
}

}  // namespace vattn
