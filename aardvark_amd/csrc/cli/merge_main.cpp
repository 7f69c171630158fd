/*
 * merge_main.cpp — `aardvark_amd_merge`: the reference's `aardvark merge` flow (src/main.rs:329-546) on top of the two C-ABIs:
 * libaardvark_feeder.so turns FASTA + BED + k VCFs into MultiRegion batches, libaardvark_amd.so runs the all-pairs exact-match
 * test on the GPU and classifies the regions, the feeder library writes passing.vcf.gz, regions.bed.gz, failed_regions.bed.gz
 * (+ .tbi each) and the summary table.  Option names are the reference's (src/cli/merge.rs).
 */
#include <cerrno>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <sys/stat.h>

#include "../../../include/aardvark_amd.h"
#include "../../../include/aardvark_feeder.h"

namespace {

double seconds_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

[[noreturn]] void die(int code, const char *what, const char *detail) {
    fprintf(stderr, "error: %s%s%s\n", what, detail && *detail ? ": " : "", detail ? detail : "");
    exit(code);
}

void usage() {
    fprintf(stderr,
            "usage: aardvark_amd_merge -r REF.fa[.gz] -i VCF [-i VCF ...] -b REGIONS.bed[.gz] -o OUT_VCF_DIR\n"
            "  [-s SAMPLE ...] [-t TAG ...] [--output-summary SUMMARY.tsv|.csv] [--output-debug DIR]\n"
            "  [--min-variant-gap 50] [--disable-variant-trimming] [--merge-strategy exact|no_conflict|majority|all]\n"
            "  [--enable-no-conflict] [--enable-voting] [--conflict-select INDEX] [--max-branch-factor 50]\n"
            "  [--skip N] [--take N] [--device 0 | --devices 0,1,..] [--batch-regions 1000000] [--contexts 2] [--batch-form packed|wide]\n");
}

std::string json_string(const std::string &s) {
    std::string out = "\"";
    for (unsigned char c : s) {
        if (c == '"') out += "\\\"";
        else if (c == '\\') out += "\\\\";
        else if (c == '\n') out += "\\n";
        else if (c == '\t') out += "\\t";
        else if (c == '\r') out += "\\r";
        else if (c < 0x20) {
            char buf[8];
            snprintf(buf, sizeof(buf), "\\u%04x", c);
            out += buf;
        } else out.push_back((char)c);
    }
    return out + "\"";
}

std::string json_list(const std::vector<std::string> &items) {
    if (items.empty()) return "[]";
    std::string out = "[\n";
    for (size_t i = 0; i < items.size(); ++i) out += "    " + json_string(items[i]) + (i + 1 < items.size() ? ",\n" : "\n");
    return out + "  ]";
}

} // namespace

int main(int argc, char **argv) {
    setenv("GPU_MAX_HW_QUEUES", "24", 0); /* before the first HIP call (see compare_main.cpp) */
    const auto t_start = std::chrono::steady_clock::now();
    std::string ref, bed, out_dir, summary_path, debug_dir, strategy;
    std::vector<std::string> vcfs, samples, tags;
    uint64_t gap = 50, branch = 50, skip = 0, take = 0, batch_regions = 1000000, threads = 1, verbosity = 0, contexts = 2;
    bool want_packed = true;
    bool trimming = true, no_conflict = false, voting = false;
    bool ref_upper = true; /* --reference-case upper|raw (include/aardvark_feeder.h, avf_genome_load_case) */
    long long conflict_select = -1;
    int device = 0;
    std::vector<int> devices; /* --devices: one solver context per entry; the regions are sharded over them by avk_region_shard (the first entry is `device`) */
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) die(64, "missing value for", a.c_str());
            return argv[++i];
        };
        if (a == "-r" || a == "--reference") ref = val();
        else if (a == "-i" || a == "--input-vcf") vcfs.push_back(val());
        else if (a == "-s" || a == "--vcf-sample") samples.push_back(val());
        else if (a == "-t" || a == "--vcf-tag") tags.push_back(val());
        else if (a == "-b" || a == "--regions") bed = val();
        else if (a == "-o" || a == "--output-vcfs") out_dir = val();
        else if (a == "--output-summary") summary_path = val();
        else if (a == "--output-debug") debug_dir = val();
        else if (a == "--min-variant-gap") gap = strtoull(val(), nullptr, 10);
        else if (a == "--disable-variant-trimming") trimming = false;
        else if (a == "--reference-case") {
            const std::string v = val();
            if (v != "upper" && v != "raw") die(78, "--reference-case must be 'upper' or 'raw'", "");
            ref_upper = v == "upper";
        }
        else if (a == "--merge-strategy") strategy = val();
        else if (a == "--enable-no-conflict") no_conflict = true;
        else if (a == "--enable-voting") voting = true;
        else if (a == "--conflict-select") conflict_select = strtoll(val(), nullptr, 10);
        else if (a == "--max-branch-factor") branch = strtoull(val(), nullptr, 10);
        else if (a == "--skip") skip = strtoull(val(), nullptr, 10);
        else if (a == "--take") take = strtoull(val(), nullptr, 10);
        else if (a == "--device") device = atoi(val());
        else if (a == "--devices") { /* e.g. 0,1,2,3 — or 0,0 for two contexts on one GPU */
            const std::string list = val();
            devices.clear();
            for (size_t b = 0; b <= list.size();) {
                const size_t e = list.find(',', b);
                const std::string item = list.substr(b, e == std::string::npos ? std::string::npos : e - b);
                if (item.empty()) die(64, "invalid value for --devices", list.c_str());
                devices.push_back(atoi(item.c_str()));
                if (e == std::string::npos) break;
                b = e + 1;
            }
        }
        else if (a == "--batch-regions") batch_regions = strtoull(val(), nullptr, 10);
        else if (a == "--contexts") contexts = strtoull(val(), nullptr, 10);
        else if (a == "--batch-form") { /* packed (default; wide when the call sets do not fit the form) | wide */
            const std::string v = val();
            if (v != "packed" && v != "wide") die(78, "--batch-form must be 'packed' or 'wide'", "");
            want_packed = v == "packed";
        }
        else if (a == "--threads") threads = strtoull(val(), nullptr, 10);
        else if (a == "-v" || a == "--verbose") verbosity += 1;
        else if (a == "-h" || a == "--help") {
            usage();
            return 0;
        } else die(64, "unknown option", a.c_str());
    }
    if (ref.empty() || vcfs.empty() || out_dir.empty()) {
        usage();
        return 64;
    }
    /* check_merge_settings (src/cli/merge.rs:178-266) */
    if (vcfs.size() > 64) die(78, "Error while verifying settings", "at most 64 input VCFs are supported");
    if (gap == 0) die(78, "Error while verifying settings", "--min-variant-gap must be >0");
    if (branch == 0 || branch > 0xFFFFFFFFull) die(78, "Error while verifying settings", "--max-branch-factor must be >0");
    std::string strategy_name; /* serde name of MergeStrategy for cli_settings.json */
    if (!strategy.empty()) {
        std::string s = strategy;
        for (char &c : s) c = (char)tolower((unsigned char)c);
        if (s == "exact") strategy_name = "Exact";
        else if (s == "no_conflict") {
            no_conflict = true;
            strategy_name = "NoConflict";
        } else if (s == "majority") {
            voting = true;
            strategy_name = "MajorityVote";
        } else if (s == "all") {
            no_conflict = voting = true;
            strategy_name = "AllOptions";
        } else die(64, "invalid value for --merge-strategy (exact, no_conflict, majority, all)", strategy.c_str());
    }
    if (conflict_select >= (long long)vcfs.size()) die(78, "Error while verifying settings", "--conflict-selection index is greater than number of provided VCFs");
    if (conflict_select < -1) die(64, "invalid value for --conflict-select", "");
    for (size_t i = tags.size(); i < vcfs.size(); ++i) tags.push_back("vcf_" + std::to_string(i));
    if (samples.size() > vcfs.size()) samples.resize(vcfs.size());
    if (batch_regions == 0) batch_regions = 1;
    if (threads == 0) threads = 1;
    if (!devices.empty()) device = devices[0];

    /* the reference genome, the call sets and the GPU context come up side by side */
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t k = (uint32_t)vcfs.size();
    std::vector<const char *> sample_ptrs(k), tag_ptrs(k);
    for (uint32_t i = 0; i < k; ++i) {
        sample_ptrs[i] = i < samples.size() ? samples[i].c_str() : "";
        tag_ptrs[i] = tags[i].c_str();
    }
    avf_genome *genome = nullptr;
    std::vector<avf_calls *> calls(k, nullptr);
    std::vector<int> rc_calls(k, 0);
    std::vector<std::string> err_calls(k);
    avk_ctx *ctx = nullptr;
    int rc_ctx = 0;
    std::string err_ctx;
    std::atomic<uint32_t> next_vcf{0};
    auto load_calls = [&] {
        for (uint32_t i = next_vcf.fetch_add(1); i < k; i = next_vcf.fetch_add(1)) {
            rc_calls[i] = avf_calls_load(vcfs[i].c_str(), sample_ptrs[i], trimming ? 1 : 0, &calls[i]);
            if (rc_calls[i]) err_calls[i] = avf_last_error(); /* the error text is per thread */
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < (k < 8 ? k : 8u); ++t) pool.emplace_back(load_calls);
    pool.emplace_back([&] {
        rc_ctx = avk_ctx_create(device, &ctx);
        if (rc_ctx) err_ctx = avk_last_error(nullptr);
    });
    const int rc_genome = avf_genome_load_case(ref.c_str(), ref_upper ? 1 : 0, &genome);
    const std::string err_genome = rc_genome ? avf_last_error() : "";
    const double s_genome = seconds_since(t0);
    for (std::thread &t : pool) t.join();
    if (rc_genome) die(74, "Error while loading reference genome", err_genome.c_str());
    for (uint32_t i = 0; i < k; ++i)
        if (rc_calls[i]) die(74, "Error while building region iterator", err_calls[i].c_str());
    if (rc_ctx) die(70, "cannot create the GPU context", err_ctx.c_str());
    const double s_load = seconds_since(t0);

    /* the reference goes to the GPU (upload + 2-bit packing) while the regions are walked */
    t0 = std::chrono::steady_clock::now();
    int rc_ref = 0;
    std::string err_ref;
    double s_ref = 0;
    std::thread th_ref([&] {
        const auto t = std::chrono::steady_clock::now();
        const uint32_t n_contigs = avf_genome_n_contigs(genome);
        std::vector<const uint8_t *> seqs(n_contigs);
        std::vector<uint64_t> lens(n_contigs);
        for (uint32_t c = 0; c < n_contigs; ++c) {
            seqs[c] = avf_genome_seq(genome, c);
            lens[c] = avf_genome_len(genome, c);
        }
        rc_ref = avk_ref_upload(ctx, n_contigs, seqs.data(), lens.data());
        if (rc_ref) err_ref = avk_last_error(ctx);
        s_ref = seconds_since(t);
    });
    avf_feed *feed = nullptr;
    const int rc_feed = avf_feed_from_calls(k, calls.data(), bed.c_str(), genome, gap, 1, &feed);
    const std::string err_feed = rc_feed ? avf_last_error() : "";
    th_ref.join();
    if (rc_feed) die(74, "Error while building region iterator", err_feed.c_str());
    if (rc_ref) die(70, "reference upload failed", err_ref.c_str());
    std::thread th_free([&calls] { /* millions of small records: released beside the solve stage */
        for (avf_calls *c : calls) avf_calls_free(c);
    });
    const avk_multi_batch *all = avf_feed_multi_batch(feed);
    const double s_feed = seconds_since(t0);
    for (uint32_t i = 0; i < k; ++i)
        fprintf(stderr, "Loaded %llu variants from input #%u.\n", (unsigned long long)avf_feed_loaded_variants(feed, (int)i), i);
    fprintf(stderr, "Found %llu segments.\n", (unsigned long long)all->n_regions);

    if (!debug_dir.empty()) { /* the CLI options as the reference saves them (src/main.rs:363-381) */
        if (mkdir(debug_dir.c_str(), 0777) != 0 && errno != EEXIST) die(74, "Error while creating debug folder", debug_dir.c_str());
        std::string js = "{\n";
        js += "  \"aardvark_version\": " + json_string(avk_version()) + ",\n";
        js += "  \"reference_fn\": " + json_string(ref) + ",\n";
        js += "  \"vcf_filenames\": " + json_list(vcfs) + ",\n";
        std::vector<std::string> sample_names(k);
        for (uint32_t i = 0; i < k; ++i) { /* the first sample of the file when none was named (src/cli/merge.rs:196-198) */
            sample_names[i] = sample_ptrs[i];
            char first[4096];
            if (sample_names[i].empty() && avf_vcf_sample_name(vcfs[i].c_str(), 0, first, sizeof(first)) == 0) sample_names[i] = first;
        }
        js += "  \"vcf_samples\": " + json_list(sample_names) + ",\n";
        js += "  \"vcf_tags\": " + json_list(tags) + ",\n";
        js += "  \"merge_regions\": " + (bed.empty() ? std::string("null") : json_string(bed)) + ",\n";
        js += "  \"output_vcf_folder\": " + json_string(out_dir) + ",\n";
        js += "  \"output_summary_filename\": " + (summary_path.empty() ? std::string("null") : json_string(summary_path)) + ",\n";
        js += "  \"debug_folder\": " + json_string(debug_dir) + ",\n";
        js += "  \"min_variant_gap\": " + std::to_string(gap) + ",\n";
        js += std::string("  \"disable_variant_trimming\": ") + (trimming ? "false" : "true") + ",\n";
        js += "  \"merge_strategy\": " + (strategy_name.empty() ? std::string("null") : json_string(strategy_name)) + ",\n";
        js += std::string("  \"enable_no_conflict\": ") + (no_conflict ? "true" : "false") + ",\n";
        js += std::string("  \"enable_voting\": ") + (voting ? "true" : "false") + ",\n";
        js += "  \"conflict_selection\": " + (conflict_select < 0 ? std::string("null") : std::to_string(conflict_select)) + ",\n";
        js += "  \"max_branch_factor\": " + std::to_string(branch) + ",\n";
        js += "  \"threads\": " + std::to_string(threads) + ",\n";
        js += "  \"verbosity\": " + std::to_string(verbosity) + ",\n";
        js += "  \"skip_blocks\": " + std::to_string(skip) + ",\n";
        js += "  \"take_blocks\": " + (take ? std::to_string(take) : std::string("18446744073709551615")) + "\n}";
        FILE *fp = fopen((debug_dir + "/cli_settings.json").c_str(), "wb");
        if (!fp || fwrite(js.data(), 1, js.size(), fp) != js.size() || fclose(fp) != 0) die(74, "Error while saving CLI options", debug_dir.c_str());
    }

    /* --skip / --take select regions by position in the iterator (src/main.rs:409-443) */
    const uint64_t first = skip < all->n_regions ? skip : all->n_regions;
    uint64_t count = all->n_regions - first;
    if (take && take < count) count = take;

    t0 = std::chrono::steady_clock::now();
    avk_merge_config cfg;
    cfg.max_branch_factor = (uint32_t)branch;
    cfg.no_conflict_enabled = no_conflict ? 1 : 0;
    cfg.majority_voting_enabled = voting ? 1 : 0;
    cfg.conflict_selection = (int32_t)conflict_select;
    /* the feed in the library's packed form (8 + k bytes per region, 5 per call + allele bytes over PCIe instead of the wide arrays); the wide ones stay for the writers */
    avk_packed_multi_batch packed_all;
    bool packed = false;
    if (want_packed) {
        const int rc_pack = avf_feed_pack_multi(feed, [](void *, size_t bytes) { return malloc(bytes); }, nullptr, &packed_all);
        if (rc_pack < 0) die(70, "cannot pack the region batch", avf_last_error());
        packed = rc_pack == 0;
        if (!packed && verbosity) fprintf(stderr, "The call sets do not fit the packed batch form (window, call count or allele length limits): using the wide form.\n");
    }
    std::vector<int32_t> status(all->n_regions + 1, -1); /* regions outside --skip/--take stay unsolved and unwritten */
    std::vector<uint8_t> classification(all->n_regions + 1, 0);
    std::vector<uint64_t> members(all->n_regions + 1, 0);
    uint64_t solved = 0, errors = 0;
    /* --devices a,b,..: the merge regions are mapped exactly like compare regions (src/main.rs:463-478), so the job is cut by the library's ONE rule, shard =
     * hash(region_id) % ranks (avk_region_shard), every context solves its shard of the packed batch and scatters status / classification / members back for the
     * writers; the only state across regions, MergeSummaryWriter's (reason, type, input) -> (pass, fail) map (src/writers/merge_summary.rs:12-18), is a dense block
     * of sums per rank (avk_merge_counts) added up by one RCCL all-reduce (avk_counts_allreduce) when every context has a GPU of its own, on the host when entries
     * repeat (RCCL does not take two ranks on one device) — and the summary table is written from those sums. */
    const size_t n_ranks = devices.size() >= 2 && packed ? devices.size() : 1;
    const uint64_t counts_len = avk_merge_counts_len(k);
    std::vector<uint64_t> job_counts;
    if (n_ranks > 1) {
        avk_packed_multi_batch sel;
        if (avf_packed_multi_slice(feed, &packed_all, first, count, &sel)) die(70, "cannot select the regions", avf_last_error());
        bool distinct = counts_len != 0;
        for (size_t i = 0; i < n_ranks; ++i)
            for (size_t j = i + 1; j < n_ranks; ++j) distinct = distinct && devices[i] != devices[j];
        std::vector<void *> comms(n_ranks, nullptr);
        void *rccl = nullptr;
        if (distinct) { /* one communicator per context (ncclCommInitAll of RCCL, looked up at run time: the tool does not link it) */
            typedef int (*init_all_fn)(void **, int, const int *);
            rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            init_all_fn init_all = rccl ? (init_all_fn)dlsym(rccl, "ncclCommInitAll") : nullptr;
            if (!init_all || init_all(comms.data(), (int)n_ranks, devices.data()) != 0) {
                if (verbosity) fprintf(stderr, "RCCL is not available (%s): the ranks' summary counters are summed on the host.\n", rccl ? "ncclCommInitAll failed" : "librccl.so not found");
                distinct = false;
                std::fill(comms.begin(), comms.end(), nullptr);
            }
        }
        std::vector<std::string> rank_err(n_ranks);
        std::vector<avk_ctx *> r_ctx(n_ranks, nullptr);
        r_ctx[0] = ctx;
        std::vector<std::vector<uint64_t>> r_counts(n_ranks, std::vector<uint64_t>(counts_len ? counts_len : 1, 0));
        auto rank_worker = [&](size_t w) {
            avk_ctx *my = ctx;
            if (w > 0) {
                my = nullptr;
                if (avk_ctx_create(devices[w], &my)) {
                    rank_err[w] = std::string("cannot create the GPU context: ") + avk_last_error(nullptr);
                    return;
                }
                r_ctx[w] = my;
                const uint32_t n_contigs = avf_genome_n_contigs(genome);
                std::vector<const uint8_t *> seqs(n_contigs);
                std::vector<uint64_t> lens(n_contigs);
                for (uint32_t c = 0; c < n_contigs; ++c) seqs[c] = avf_genome_seq(genome, c), lens[c] = avf_genome_len(genome, c);
                if (avk_ref_upload(my, n_contigs, seqs.data(), lens.data())) { /* the reference is replicated */
                    rank_err[w] = std::string("reference upload failed: ") + avk_last_error(my);
                    return;
                }
            }
            avk_packed_multi_shard *shard = nullptr;
            if (avk_packed_multi_shard_make(&sel, all->region_id + first, 0, (uint32_t)w, (uint32_t)n_ranks, &shard)) {
                rank_err[w] = "cannot cut the shard";
                return;
            }
            const avk_packed_multi_batch *sb = avk_packed_multi_shard_batch(shard);
            std::vector<int32_t> s_status(sb->n_regions + 1);
            std::vector<uint8_t> s_cls(sb->n_regions + 1);
            std::vector<uint64_t> s_members(sb->n_regions + 1);
            if (avk_merge_packed(my, sb, &cfg, s_status.data(), s_cls.data(), s_members.data())) rank_err[w] = std::string("merge failed: ") + avk_last_error(my);
            else {
                (void)avk_packed_multi_shard_scatter(shard, s_status.data(), s_cls.data(), s_members.data(), status.data() + first, classification.data() + first,
                                                     members.data() + first);
                if (counts_len && avk_merge_counts(sb, s_status.data(), s_cls.data(), s_members.data(), r_counts[w].data())) rank_err[w] = "cannot count the shard's variants";
            }
            avk_packed_multi_shard_free(shard);
        };
        std::vector<std::thread> rank_pool;
        for (size_t w = 1; w < n_ranks; ++w) rank_pool.emplace_back(rank_worker, w);
        rank_worker(0);
        for (std::thread &t : rank_pool) t.join();
        rank_pool.clear();
        bool all_solved = true;
        for (size_t w = 0; w < n_ranks; ++w) all_solved = all_solved && rank_err[w].empty();
        /* the collective: entered only when every rank has its sums, so a rank that failed never leaves the others waiting inside ncclAllReduce */
        if (all_solved && distinct) {
            std::vector<std::vector<uint64_t>> reduced = r_counts;
            std::vector<std::string> reduce_err(n_ranks);
            auto reduce_worker = [&](size_t w) {
                if (avk_counts_allreduce(r_ctx[w], comms[w], reduced[w].data(), counts_len)) reduce_err[w] = avk_last_error(r_ctx[w]);
            };
            for (size_t w = 1; w < n_ranks; ++w) rank_pool.emplace_back(reduce_worker, w);
            reduce_worker(0);
            for (std::thread &t : rank_pool) t.join();
            for (size_t w = 0; w < n_ranks; ++w)
                if (!reduce_err[w].empty()) {
                    if (verbosity) fprintf(stderr, "all-reduce of the summary counters failed (%s): they are summed on the host.\n", reduce_err[w].c_str());
                    distinct = false;
                    break;
                }
            if (distinct) job_counts = reduced[0]; /* every rank holds the job's sums */
        } else distinct = false;
        for (size_t w = 0; w < n_ranks; ++w) {
            if (comms[w] && rccl) {
                typedef int (*destroy_fn)(void *);
                destroy_fn destroy = (destroy_fn)dlsym(rccl, "ncclCommDestroy");
                if (destroy) (void)destroy(comms[w]);
            }
            if (w > 0 && r_ctx[w]) avk_ctx_destroy(r_ctx[w]);
        }
        for (size_t w = 0; w < n_ranks; ++w)
            if (!rank_err[w].empty()) die(70, rank_err[w].c_str(), "");
        if (!distinct && counts_len) {
            job_counts.assign(counts_len, 0);
            for (size_t w = 0; w < n_ranks; ++w)
                for (uint64_t i = 0; i < counts_len; ++i) job_counts[i] += r_counts[w][i];
        }
        if (verbosity)
            fprintf(stderr, "%zu contexts, regions sharded by hash(region_id) %% %zu; the summary counters summed %s.\n", n_ranks, n_ranks,
                    distinct ? "by one RCCL all-reduce" : counts_len ? "on the host" : "from the scattered per-region results (more inputs than the dense counters take)");
    }
    /* Two contexts on the same GPU work through the batches: the host side of one batch (pair records, packing, upload, classification)
     * runs beside the kernels of the other.  The second context is made by its own thread, reference upload included. */
    const uint64_t n_batches = (count + batch_regions - 1) / batch_regions;
    const int n_workers = n_ranks > 1 ? 0 : n_batches >= 2 && contexts >= 2 ? 2 : 1;
    std::atomic<uint64_t> next_batch{0};
    std::string worker_err[2];
    auto solve_batches = [&](int w) {
        avk_ctx *my = ctx;
        if (w > 0) {
            my = nullptr;
            if (avk_ctx_create(device, &my)) {
                worker_err[w] = std::string("cannot create the GPU context: ") + avk_last_error(nullptr);
                return;
            }
            const uint32_t n_contigs = avf_genome_n_contigs(genome);
            std::vector<const uint8_t *> seqs(n_contigs);
            std::vector<uint64_t> lens(n_contigs);
            for (uint32_t c = 0; c < n_contigs; ++c) {
                seqs[c] = avf_genome_seq(genome, c);
                lens[c] = avf_genome_len(genome, c);
            }
            if (avk_ref_upload(my, n_contigs, seqs.data(), lens.data())) {
                worker_err[w] = std::string("reference upload failed: ") + avk_last_error(my);
                avk_ctx_destroy(my);
                return;
            }
        }
        for (uint64_t bi = next_batch.fetch_add(1); bi < n_batches; bi = next_batch.fetch_add(1)) {
            const uint64_t at = bi * batch_regions;
            const uint64_t n = count - at < batch_regions ? count - at : batch_regions;
            avk_multi_batch b = *all; /* a window of the region arrays; variant arrays are shared */
            b.n_regions = n;
            b.region_id = all->region_id + first + at;
            b.contig_idx = all->contig_idx + first + at;
            b.start = all->start + first + at;
            b.end = all->end + first + at;
            b.in_off = all->in_off + (first + at) * k;
            b.in_cnt = all->in_cnt + (first + at) * k;
            int rc_merge = 0;
            if (packed) {
                avk_packed_multi_batch part;
                rc_merge = avf_packed_multi_slice(feed, &packed_all, first + at, n, &part);
                if (!rc_merge) rc_merge = avk_merge_packed(my, &part, &cfg, status.data() + first + at, classification.data() + first + at, members.data() + first + at);
            } else
                rc_merge = avk_merge_batch(my, &b, &cfg, status.data() + first + at, classification.data() + first + at, members.data() + first + at);
            if (rc_merge) {
                worker_err[w] = std::string("merge failed: ") + avk_last_error(my);
                break;
            }
        }
        if (w > 0) avk_ctx_destroy(my);
    };
    {
        std::thread other;
        if (n_workers == 2) other = std::thread(solve_batches, 1);
        if (n_workers >= 1) solve_batches(0);
        if (other.joinable()) other.join();
        for (int w = 0; w < 2; ++w)
            if (!worker_err[w].empty()) die(70, worker_err[w].c_str(), "");
    }
    for (uint64_t r = first; r < first + count; ++r) {
        if (status[r] == 0) {
            solved += 1;
            continue;
        }
        errors += 1;
        fprintf(stderr, "Error while solving merge region #%llu (contig %u:%llu-%llu): status %d\n", (unsigned long long)all->region_id[r], all->contig_idx[r],
                (unsigned long long)all->start[r], (unsigned long long)all->end[r], status[r]);
    }
    const double s_solve = seconds_since(t0);

    t0 = std::chrono::steady_clock::now();
    std::string command;
    for (int i = 0; i < argc; ++i) command += (i ? " " : "") + std::string(argv[i]);
    if (avf_write_merge_outputs(out_dir.c_str(), vcfs[0].c_str(), sample_ptrs[0], avk_version(), command.c_str(), genome, all, tag_ptrs.data(), status.data(),
                                classification.data(), members.data()))
        die(74, "Error while writing merged VCF results", avf_last_error());
    fprintf(stderr, "Solved:error blocks: %llu : %llu\n", (unsigned long long)solved, (unsigned long long)errors);
    if (!summary_path.empty() &&
        (job_counts.empty() ? avf_write_merge_summary(summary_path.c_str(), all, tag_ptrs.data(), status.data(), classification.data(), members.data())
                            : avf_write_merge_summary_counts(summary_path.c_str(), k, tag_ptrs.data(), job_counts.data(), job_counts.size())))
        die(74, "Error while saving summary file", avf_last_error());
    const double s_write = seconds_since(t0);

    fprintf(stderr, "stages [s]: load %.3f (reference %.3f beside the call sets and the gpu context), regions %.3f (beside it: reference upload %.3f), "
                    "solve (pack + H2D + kernels + D2H + classify) %.3f, outputs %.3f\n",
            s_load, s_genome, s_feed, s_ref, s_solve, s_write);
    fprintf(stderr, "Merge completed in %.3f seconds.\n", seconds_since(t_start));
    th_free.join();
    avk_ctx_destroy(ctx);
    avf_feed_free(feed);
    avf_genome_free(genome);
    return 0;
}
