/*
 * compare_main.cpp — `aardvark_amd_compare`: the reference's `aardvark compare` flow (src/main.rs:30-327) on top of the
 * two C-ABIs: libaardvark_feeder.so turns FASTA + BED + truth/query VCFs into region batches, libaardvark_amd.so
 * solves them on the GPU, the feeder library writes summary.tsv.  Option names are the reference's
 * (src/cli/compare.rs), --stratification included.  Outputs: summary.tsv, truth.vcf.gz, query.vcf.gz (+ .tbi).
 * --output-debug writes cli_settings.json, region_summary.tsv.gz and region_sequences.tsv.gz.
 */
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

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
            "usage: aardvark_amd_compare -r REF.fa[.gz] -t TRUTH.vcf[.gz] -q QUERY.vcf[.gz] -b REGIONS.bed[.gz] -o OUT_DIR\n"
            "  [--truth-sample S] [--query-sample S] [--compare-label L] [--min-variant-gap 50] [--disable-variant-trimming]\n"
            "  [--reference-case upper|raw]  (default upper: soft-masked reference bases are compared as upper case)\n"
            "  [--max-branch-factor 50] [--enable-exact-shortcut] [--enable-haplotype-metrics] [--enable-weighted-haplotype-metrics]\n"
            "  [--enable-record-basepair-metrics] [-s STRAT.tsv] [--output-debug DIR] [--skip N] [--take N] [--device 0 | --devices 0,1,..] [--batch-regions 4000000] [--batch-form packed|wide]\n");
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

} // namespace

/* One line per region that was not solved.  A region this build REFUSES for its size (more than 60,000 calls, a window of 2^31 bases or
 * more — limits the reference does not have) says so, so that it is not taken for an error the reference would
 * report too. */
static void report_unsolved(const avk_region_batch &b, uint64_t r, int32_t status) {
    const uint64_t calls = (uint64_t)b.t_cnt[r] + b.q_cnt[r], window = b.end[r] - b.start[r];
    if (status == AVK_ST_INVALID_INPUT && (calls > 60000 || window >= (1ull << 31)))
        fprintf(stderr, "Region #%llu (contig %u:%llu-%llu) NOT COMPARED: %llu calls in a window of %llu bases is beyond this build's limits (60,000 calls, 2^31 bases); the reference has no such limit\n",
                (unsigned long long)b.region_id[r], b.contig_idx[r], (unsigned long long)b.start[r], (unsigned long long)b.end[r], (unsigned long long)calls, (unsigned long long)window);
    else
        fprintf(stderr, "Error while solving compare region #%llu (contig %u:%llu-%llu): status %d\n", (unsigned long long)b.region_id[r], b.contig_idx[r],
                (unsigned long long)b.start[r], (unsigned long long)b.end[r], status);
}

int main(int argc, char **argv) {
    setenv("GPU_MAX_HW_QUEUES", "24", 0); /* before the first HIP call: the solver's six streams need hardware queues of their own (include/aardvark_amd.h, avk_ctx_create) */
    const auto t_start = std::chrono::steady_clock::now();
    std::string ref, truth, query, bed, out_dir, truth_sample, query_sample, label = "compare", strat_tsv, debug_dir;
    uint64_t gap = 50, branch = 50, skip = 0, take = 0, batch_regions = 4000000, threads = 1, max_ed = 5000, verbosity = 0;
    bool trimming = true, shortcut = false, hap = false, whap = false, rbp = false;
    bool ref_upper = true; /* --reference-case upper|raw (include/aardvark_feeder.h, avf_genome_load_case) */
    int device = 0;
    std::vector<int> devices; /* --devices: the contexts that share the region batches (the first one is `device`) */
    bool batch_given = false, want_packed = true;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) die(64, "missing value for", a.c_str());
            return argv[++i];
        };
        if (a == "-r" || a == "--reference") ref = val();
        else if (a == "-t" || a == "--truth-vcf") truth = val();
        else if (a == "-q" || a == "--query-vcf") query = val();
        else if (a == "-b" || a == "--regions") bed = val();
        else if (a == "-o" || a == "--output-dir") out_dir = val();
        else if (a == "--truth-sample") truth_sample = val();
        else if (a == "--query-sample") query_sample = val();
        else if (a == "--compare-label") label = val();
        else if (a == "--min-variant-gap") gap = strtoull(val(), nullptr, 10);
        else if (a == "--disable-variant-trimming") trimming = false;
        else if (a == "--reference-case") {
            const std::string v = val();
            if (v != "upper" && v != "raw") die(78, "--reference-case must be 'upper' or 'raw'", "");
            ref_upper = v == "upper";
        }
        else if (a == "--max-branch-factor") branch = strtoull(val(), nullptr, 10);
        else if (a == "--enable-exact-shortcut") shortcut = true;
        else if (a == "--enable-haplotype-metrics") hap = true;
        else if (a == "--enable-weighted-haplotype-metrics") whap = true;
        else if (a == "--enable-record-basepair-metrics") rbp = true;
        else if (a == "--skip") skip = strtoull(val(), nullptr, 10);
        else if (a == "--take") take = strtoull(val(), nullptr, 10);
        else if (a == "--device") device = atoi(val());
        else if (a == "--batch-regions") {
            batch_regions = strtoull(val(), nullptr, 10);
            batch_given = true;
        } else if (a == "--devices") { /* one solver context per entry, e.g. 0,1,2,3 — or 0,0 for two contexts on one GPU */
            std::string list = val();
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
        else if (a == "--batch-form") { /* packed (default; wide when the call set does not fit the form, or with --output-debug) | wide */
            const std::string v = val();
            if (v != "packed" && v != "wide") die(78, "--batch-form must be 'packed' or 'wide'", "");
            want_packed = v == "packed";
        }
        else if (a == "--threads") threads = strtoull(val(), nullptr, 10);          /* accepted for command-line compatibility */
        else if (a == "--max-edit-distance") max_ed = strtoull(val(), nullptr, 10); /* hidden in the reference as well, unused by it */
        else if (a == "-v" || a == "--verbose") verbosity += 1;
        else if (a == "-s" || a == "--stratification") strat_tsv = val();
        else if (a == "--output-debug") debug_dir = val();
        else if (a == "-h" || a == "--help") {
            usage();
            return 0;
        } else die(64, "unknown option", a.c_str());
    }
    if (ref.empty() || truth.empty() || query.empty() || out_dir.empty()) {
        usage();
        return 64;
    }
    if (gap == 0) die(78, "--min-variant-gap must be >0", "");
    if (branch == 0 || branch > 0xFFFFFFFFull) die(78, "--max-branch-factor must be >0", "");
    if (batch_regions == 0) batch_regions = 1;
    if (!devices.empty()) device = devices[0];
    if (mkdir(out_dir.c_str(), 0777) != 0 && errno != EEXIST) die(74, "cannot create output folder", out_dir.c_str());
    if (!debug_dir.empty() && mkdir(debug_dir.c_str(), 0777) != 0 && errno != EEXIST) die(74, "cannot create debug folder", debug_dir.c_str());
    if (!debug_dir.empty()) { /* the CLI options as the reference saves them (src/main.rs:64-82; serde field order of CompareSettings) */
        auto opt = [&](const std::string &v) { return v.empty() ? std::string("null") : json_string(v); };
        auto flag = [](bool v) { return std::string(v ? "true" : "false"); };
        std::string samples[2] = {truth_sample, query_sample};
        const std::string *files[2] = {&truth, &query};
        for (int i = 0; i < 2; ++i) { /* the first sample of the file when none was named (src/cli/compare.rs:186-194) */
            char first[4096];
            if (samples[i].empty() && avf_vcf_sample_name(files[i]->c_str(), 0, first, sizeof(first)) == 0) samples[i] = first;
        }
        std::string js = "{\n";
        js += "  \"aardvark_version\": " + json_string(avk_version()) + ",\n";
        js += "  \"reference_fn\": " + json_string(ref) + ",\n";
        js += "  \"truth_vcf_filename\": " + json_string(truth) + ",\n";
        js += "  \"query_vcf_filename\": " + json_string(query) + ",\n";
        js += "  \"regions\": " + opt(bed) + ",\n";
        js += "  \"stratifications\": " + opt(strat_tsv) + ",\n";
        js += "  \"output_folder\": " + json_string(out_dir) + ",\n";
        js += "  \"debug_folder\": " + json_string(debug_dir) + ",\n";
        js += "  \"compare_label\": " + json_string(label) + ",\n";
        js += "  \"truth_sample\": " + json_string(samples[0]) + ",\n";
        js += "  \"query_sample\": " + json_string(samples[1]) + ",\n";
        js += "  \"min_variant_gap\": " + std::to_string(gap) + ",\n";
        js += "  \"disable_variant_trimming\": " + flag(!trimming) + ",\n";
        js += "  \"max_edit_distance\": " + std::to_string(max_ed) + ",\n";
        js += "  \"max_branch_factor\": " + std::to_string(branch) + ",\n";
        js += "  \"enable_exact_shortcut\": " + flag(shortcut) + ",\n";
        js += "  \"enable_haplotype_scoring\": " + flag(hap) + ",\n";
        js += "  \"enable_weighted_haplotype_scoring\": " + flag(whap) + ",\n";
        js += "  \"enable_record_basepair_scoring\": " + flag(rbp) + ",\n";
        js += "  \"threads\": " + std::to_string(threads ? threads : 1) + ",\n";
        js += "  \"verbosity\": " + std::to_string(verbosity) + ",\n";
        js += "  \"skip_blocks\": " + std::to_string(skip) + ",\n";
        js += "  \"take_blocks\": " + (take ? std::to_string(take) : std::string("18446744073709551615")) + "\n}";
        FILE *fp = fopen((debug_dir + "/cli_settings.json").c_str(), "wb");
        if (!fp || fwrite(js.data(), 1, js.size(), fp) != js.size() || fclose(fp) != 0) die(74, "Error while saving CLI options", debug_dir.c_str());
    }

    /* the reference genome, the two call sets and the GPU context come up side by side */
    auto t0 = std::chrono::steady_clock::now();
    avf_genome *genome = nullptr;
    avf_calls *calls[2] = {nullptr, nullptr};
    avk_ctx *ctx = nullptr;
    int rc_calls[2] = {0, 0}, rc_ctx = 0;
    std::string err_calls[2], err_ctx;
    double s_calls[2] = {0, 0}, s_ctx = 0;
    std::thread th_calls[2], th_ctx;
    for (int i = 0; i < 2; ++i)
        th_calls[i] = std::thread([&, i] {
            const auto t = std::chrono::steady_clock::now();
            rc_calls[i] = avf_calls_load(i == 0 ? truth.c_str() : query.c_str(), i == 0 ? truth_sample.c_str() : query_sample.c_str(), trimming ? 1 : 0, &calls[i]);
            if (rc_calls[i]) err_calls[i] = avf_last_error(); /* the error text is per thread */
            s_calls[i] = seconds_since(t);
        });
    std::thread th_reserve;
    th_ctx = std::thread([&] {
        const auto t = std::chrono::steady_clock::now();
        rc_ctx = avk_ctx_create(device, &ctx);
        if (rc_ctx) err_ctx = avk_last_error(nullptr);
        else { /* the staging buffers of the solve stage while the inputs are still being read: a first guess from the size of the reference
                  file (a region per 750 bases is twice the density of a human call set); the exact sizes follow when the calls are loaded */
            struct stat st;
            if (stat(ref.c_str(), &st) == 0 && st.st_size > (256ll << 20)) {
                const bool gz = ref.size() > 3 && ref.compare(ref.size() - 3, 3, ".gz") == 0;
                const uint64_t guess = (uint64_t)st.st_size * (gz ? 4u : 1u) / 750u;
                th_reserve = std::thread([guess, &ctx] { (void)avk_ctx_warmup(ctx, guess, 2 * guess); }); /* (bounce buffer, device code, workspaces) joined before the solve stage */
            }
        }
        s_ctx = seconds_since(t);
    });
    const int rc_genome = avf_genome_load_case(ref.c_str(), ref_upper ? 1 : 0, &genome);
    const std::string err_genome = rc_genome ? avf_last_error() : "";
    const double s_genome = seconds_since(t0);
    for (std::thread &t : th_calls) t.join();
    th_ctx.join();
    if (rc_genome) die(74, "Error while loading reference genome", err_genome.c_str());
    for (int i = 0; i < 2; ++i)
        if (rc_calls[i]) die(74, "Error while building regions", err_calls[i].c_str());
    if (rc_ctx) die(70, "cannot create the GPU context", err_ctx.c_str());
    const double s_load = seconds_since(t0);

    avf_strat *strat = nullptr;
    if (!strat_tsv.empty() && avf_strat_load(strat_tsv.c_str(), &strat)) die(74, "Error while loading stratifications", avf_last_error());
    const uint32_t n_labels = avf_strat_n_labels(strat);

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
    const int rc_feed = avf_feed_from_calls(2, calls, bed.c_str(), genome, gap, 0, &feed);
    const std::string err_feed = rc_feed ? avf_last_error() : "";
    th_ref.join();
    if (rc_feed) die(74, "Error while building regions", err_feed.c_str());
    if (rc_ref) die(70, "reference upload failed", err_ref.c_str());
    std::thread th_free([&calls] { /* millions of small records: released beside the solve stage */
        avf_calls_free(calls[0]);
        avf_calls_free(calls[1]);
    });
    const avk_region_batch *all = avf_feed_batch(feed);
    if (th_reserve.joinable()) th_reserve.join();
    const double s_feed = seconds_since(t0);
    fprintf(stderr, "Loaded %llu truth and %llu query variants; %llu regions.\n", (unsigned long long)avf_feed_loaded_variants(feed, 0),
            (unsigned long long)avf_feed_loaded_variants(feed, 1), (unsigned long long)all->n_regions);

    /* the feed in the library's packed form (10 bytes per region, 5 per call + allele bytes over PCIe instead of 479 MB per whole genome); the wide arrays
     * stay for the writers.  Ordinary memory: pinning it would cost more than the one pass through the library's bounce buffer it saves. */
    avk_packed_batch packed_all;
    bool packed = false;
    if (want_packed && debug_dir.empty()) {
        const int rc_pack = avf_feed_pack(feed, [](void *, size_t bytes) { return malloc(bytes); }, nullptr, &packed_all);
        if (rc_pack < 0) die(70, "cannot pack the region batch", avf_last_error());
        packed = rc_pack == 0;
        if (!packed && verbosity) fprintf(stderr, "The call set does not fit the packed batch form (window, call count or allele length limits): using the wide form.\n");
    }
    /* regions [first + at, +n): as a batch in the chosen form; `out` is indexed by the feed's call arrays either way */
    auto out_for = [](avk_result_batch out, uint64_t v_first) { /* a packed part's results are indexed from its first call */
        if (out.var_expected) out.var_expected += v_first;
        if (out.var_observed) out.var_observed += v_first;
        if (out.var_class) out.var_class += v_first;
        if (out.var_zyg) out.var_zyg += v_first;
        return out;
    };
    auto compare_part = [&](avk_ctx *c, const avk_region_batch &b, uint64_t at_abs, uint64_t n, const avk_compare_config &cfg, const avk_result_batch &out) -> int {
        if (!packed) return avk_compare_batch(c, &b, &cfg, const_cast<avk_result_batch *>(&out));
        avk_packed_batch part;
        uint64_t v_first = 0;
        if (avf_packed_slice(feed, &packed_all, at_abs, n, &part, &v_first)) return AVK_E_ARG;
        avk_result_batch shifted = out_for(out, v_first);
        return avk_compare_packed(c, &part, &cfg, &shifted);
    };
    auto upload_part = [&](avk_ctx *c, const avk_region_batch &b, uint64_t at_abs, uint64_t n, avk_dev_batch **db, uint64_t *v_first) -> int {
        *v_first = 0;
        if (!packed) return avk_batch_upload(c, &b, db);
        avk_packed_batch part;
        if (avf_packed_slice(feed, &packed_all, at_abs, n, &part, v_first)) return AVK_E_ARG;
        return avk_batch_upload_packed(c, &part, db);
    };

    /* --skip / --take select regions by position in the iterator (src/main.rs:215-231) */
    const uint64_t first = skip < all->n_regions ? skip : all->n_regions;
    uint64_t count = all->n_regions - first;
    if (take && take < count) count = take;

    t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> total(AVK_TALLY_LEN, 0), tally(AVK_TALLY_LEN);
    avk_compare_config cfg;
    cfg.max_branch_factor = (uint32_t)branch;
    cfg.enable_sequences = 0;
    cfg.enable_exact_shortcut = shortcut ? 1 : 0;
    std::vector<int32_t> status(all->n_regions, -1); /* regions outside --skip/--take stay unsolved and unwritten */
    std::vector<uint8_t> var_expected(all->n_variants + 1), var_observed(all->n_variants + 1), var_class(all->n_variants + 1);
    /* stratified tallies (SummaryWriter::add_comparison_benchmark, summary.rs:146-163): label l sums the metric blocks of the
     * regions it contains, so the per-region blocks come back from the GPU, in batches that keep them at a few hundred MB */
    const bool debug = !debug_dir.empty();
    (void)avk_ctx_set_option(ctx, "emit_group_metrics", n_labels || debug ? 1 : 0);
    /* the per-region blocks only come back to the host for the debug tables; the stratified tallies are summed on the GPU
     * (avk_label_tallies) from the region labels the feeder library lists (avf_strat_batch_labels) */
    const bool device_labels = n_labels && !debug;
    if (debug && batch_regions > 262144) batch_regions = 262144;
    cfg.enable_sequences = debug ? 1 : 0; /* enable_sequences(region_seq_writer.is_some()), src/main.rs:236 */
    uint32_t mask = AVF_METRIC_GT | AVF_METRIC_BASEPAIR;
    if (hap) mask |= AVF_METRIC_HAP;
    if (whap) mask |= AVF_METRIC_WEIGHTED_HAP;
    if (rbp) mask |= AVF_METRIC_RECORD_BP;
    avf_table *region_table = nullptr, *sequence_table = nullptr;
    if (debug) { /* the debug tables (src/main.rs:163-188) */
        if (avf_region_summary_open((debug_dir + "/region_summary.tsv.gz").c_str(), mask, &region_table) ||
            avf_region_sequences_open((debug_dir + "/region_sequences.tsv.gz").c_str(), &sequence_table))
            die(74, "Error while building debug writers", avf_last_error());
    }
    std::vector<uint8_t> seq_bytes;
    std::vector<uint64_t> seq_off;
    std::vector<uint32_t> seq_stride, seq_len;
    std::vector<uint64_t> strat_total((size_t)n_labels * AVK_TALLY_LEN, 0);
    std::vector<uint32_t> gm, labels(n_labels ? n_labels : 1);
    /* Several contexts (--devices: one per entry, GPUs may repeat) share the region batches: regions are independent, every context
     * solves the batches it draws, the tallies are summed on the host (the sum the benchmark does with one RCCL all-reduce across
     * processes).  The debug tables are written in region order by one context. */
    const size_t n_workers = debug || devices.size() < 2 ? 1 : devices.size();
    /* --devices with the packed feed and no stratification: the job is cut by the library's ONE rule, shard = hash(region_id) % ranks (avk_region_shard, the rule of
     * aardvark_amd/dist.py), every context solves its shard, and the job tally is summed over the ranks — by one RCCL all-reduce (avk_tally_allreduce) when every
     * context has a GPU of its own, on the host when entries repeat (RCCL does not take two ranks on one device). */
    bool sharded = false;
    if (n_workers > 1 && packed && !n_labels) {
        sharded = true;
        avk_packed_batch sel;
        uint64_t sel_v = 0;
        if (avf_packed_slice(feed, &packed_all, first, count, &sel, &sel_v)) die(70, "cannot select the regions", avf_last_error());
        bool distinct = true;
        for (size_t i = 0; i < n_workers; ++i)
            for (size_t j = i + 1; j < n_workers; ++j) distinct = distinct && devices[i] != devices[j];
        std::vector<void *> comms(n_workers, nullptr);
        if (distinct) { /* one communicator per context, made here (ncclCommInitAll of RCCL, looked up at run time: the tool does not link it) */
            typedef int (*init_all_fn)(void **, int, const int *);
            void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            init_all_fn init_all = h ? (init_all_fn)dlsym(h, "ncclCommInitAll") : nullptr;
            if (!init_all || init_all(comms.data(), (int)n_workers, devices.data()) != 0) {
                if (verbosity) fprintf(stderr, "RCCL is not available (%s): the ranks' tallies are summed on the host.\n", h ? "ncclCommInitAll failed" : "librccl.so not found");
                distinct = false;
                std::fill(comms.begin(), comms.end(), nullptr);
            }
        }
        std::vector<std::string> worker_err(n_workers);
        std::vector<std::vector<uint64_t>> w_total(n_workers, std::vector<uint64_t>(AVK_TALLY_LEN, 0));
        std::mutex log_mutex;
        std::vector<avk_ctx *> w_ctx(n_workers, nullptr); /* the contexts outlive the workers: the collective runs after every rank is known to have solved its shard */
        w_ctx[0] = ctx;
        auto shard_worker = [&](size_t w) {
            avk_ctx *my = ctx;
            if (w > 0) {
                my = nullptr;
                if (avk_ctx_create(devices[w], &my)) {
                    worker_err[w] = std::string("cannot create the GPU context: ") + avk_last_error(nullptr);
                    return;
                }
                w_ctx[w] = my;
                const uint32_t n_contigs = avf_genome_n_contigs(genome);
                std::vector<const uint8_t *> seqs(n_contigs);
                std::vector<uint64_t> lens(n_contigs);
                for (uint32_t c = 0; c < n_contigs; ++c) seqs[c] = avf_genome_seq(genome, c), lens[c] = avf_genome_len(genome, c);
                if (avk_ref_upload(my, n_contigs, seqs.data(), lens.data())) { /* the reference is replicated: 0.8 GB packed per rank */
                    worker_err[w] = std::string("reference upload failed: ") + avk_last_error(my);
                    return;
                }
                (void)avk_ctx_set_option(my, "emit_group_metrics", 0);
            }
            avk_packed_shard *shard = nullptr;
            if (avk_packed_shard_make(&sel, all->region_id + first, 0, (uint32_t)w, (uint32_t)n_workers, &shard)) worker_err[w] = "cannot cut the shard";
            else {
                const avk_packed_batch *sb = avk_packed_shard_batch(shard);
                std::vector<int32_t> s_status(sb->n_regions + 1);
                std::vector<uint8_t> s_e(sb->n_variants + 1), s_o(sb->n_variants + 1), s_c(sb->n_variants + 1);
                avk_result_batch so;
                memset(&so, 0, sizeof(so));
                so.status = s_status.data(), so.var_expected = s_e.data(), so.var_observed = s_o.data(), so.var_class = s_c.data(), so.tally = w_total[w].data();
                if (avk_compare_packed(my, sb, &cfg, &so)) worker_err[w] = std::string("compare failed: ") + avk_last_error(my);
                else {
                    avk_result_batch to; /* the selected regions' part of the job's arrays: regions from `first`, calls from the selection's first call */
                    memset(&to, 0, sizeof(to));
                    to.status = status.data() + first, to.var_expected = var_expected.data() + sel_v, to.var_observed = var_observed.data() + sel_v, to.var_class = var_class.data() + sel_v;
                    (void)avk_packed_shard_scatter(shard, &so, &to);
                    const uint64_t *idx = nullptr;
                    const uint64_t m = avk_packed_shard_regions(shard, &idx);
                    for (uint64_t k = 0; k < m; ++k)
                        if (s_status[k] != 0) {
                            std::lock_guard<std::mutex> lock(log_mutex);
                            avk_region_batch b1 = *all;
                            b1.region_id = all->region_id + first, b1.contig_idx = all->contig_idx + first, b1.start = all->start + first, b1.end = all->end + first;
                            b1.t_off = all->t_off + first, b1.t_cnt = all->t_cnt + first, b1.q_off = all->q_off + first, b1.q_cnt = all->q_cnt + first;
                            report_unsolved(b1, idx[k], s_status[k]);
                        }
                }
                avk_packed_shard_free(shard);
            }
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < n_workers; ++w) pool.emplace_back(shard_worker, w);
        shard_worker(0);
        for (std::thread &t : pool) t.join();
        pool.clear();
        bool all_solved = true;
        for (size_t w = 0; w < n_workers; ++w) all_solved = all_solved && worker_err[w].empty();
        /* the collective: entered only when EVERY rank has its sums — a rank that failed above never leaves the others waiting inside ncclAllReduce */
        if (all_solved && distinct) {
            std::vector<std::vector<uint64_t>> reduced = w_total;
            std::vector<std::string> reduce_err(n_workers);
            auto reduce_worker = [&](size_t w) {
                if (avk_tally_allreduce(w_ctx[w], comms[w], reduced[w].data())) reduce_err[w] = avk_last_error(w_ctx[w]);
            };
            for (size_t w = 1; w < n_workers; ++w) pool.emplace_back(reduce_worker, w);
            reduce_worker(0);
            for (std::thread &t : pool) t.join();
            for (size_t w = 0; w < n_workers; ++w)
                if (!reduce_err[w].empty()) { /* the ranks' own tallies are still there: the host sum stands in */
                    if (verbosity) fprintf(stderr, "tally all-reduce failed (%s): the ranks' tallies are summed on the host.\n", reduce_err[w].c_str());
                    distinct = false;
                    break;
                }
            if (distinct) total = reduced[0]; /* every rank holds the job's sums */
        } else if (!all_solved) distinct = false;
        for (size_t w = 0; w < n_workers; ++w) {
            if (comms[w]) {
                typedef int (*destroy_fn)(void *);
                void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
                destroy_fn destroy = h ? (destroy_fn)dlsym(h, "ncclCommDestroy") : nullptr;
                if (destroy) (void)destroy(comms[w]);
            }
            if (w > 0 && w_ctx[w]) avk_ctx_destroy(w_ctx[w]);
        }
        for (size_t w = 0; w < n_workers; ++w)
            if (!worker_err[w].empty()) die(70, worker_err[w].c_str(), "");
        if (!distinct)
            for (size_t w = 0; w < n_workers; ++w)
                for (size_t k = 0; k < (size_t)AVK_TALLY_LEN; ++k) total[k] += w_total[w][k];
        if (verbosity) fprintf(stderr, "%zu contexts, regions sharded by hash(region_id) %% %zu; the job tally summed %s.\n", n_workers, n_workers, distinct ? "by one RCCL all-reduce" : "on the host");
    }
    if (n_workers > 1 && !sharded) {
        if (!batch_given) { /* about two batches per context */
            batch_regions = (count + 2 * n_workers - 1) / (2 * n_workers);
            if (batch_regions < 100000) batch_regions = 100000;
        }
        const uint64_t n_batches = (count + batch_regions - 1) / batch_regions;
        std::atomic<uint64_t> next_batch{0};
        std::vector<std::string> worker_err(n_workers);
        std::vector<std::vector<uint64_t>> w_total(n_workers, std::vector<uint64_t>(AVK_TALLY_LEN, 0)),
            w_strat(n_workers, std::vector<uint64_t>((size_t)n_labels * AVK_TALLY_LEN, 0));
        std::mutex log_mutex;
        auto worker = [&](size_t w) {
            avk_ctx *my = ctx;
            if (w > 0) {
                my = nullptr;
                if (avk_ctx_create(devices[w], &my)) {
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
                (void)avk_ctx_set_option(my, "emit_group_metrics", n_labels ? 1 : 0);
            }
            std::vector<uint64_t> w_tally(AVK_TALLY_LEN);
            for (uint64_t bi = next_batch.fetch_add(1); bi < n_batches; bi = next_batch.fetch_add(1)) {
                const uint64_t at = bi * batch_regions;
                const uint64_t n = count - at < batch_regions ? count - at : batch_regions;
                avk_region_batch b = *all;
                b.n_regions = n;
                b.region_id = all->region_id + first + at;
                b.contig_idx = all->contig_idx + first + at;
                b.start = all->start + first + at;
                b.end = all->end + first + at;
                b.t_off = all->t_off + first + at;
                b.t_cnt = all->t_cnt + first + at;
                b.q_off = all->q_off + first + at;
                b.q_cnt = all->q_cnt + first + at;
                avk_result_batch out;
                memset(&out, 0, sizeof(out));
                out.status = status.data() + first + at;
                out.tally = w_tally.data();
                out.var_expected = var_expected.data();
                out.var_observed = var_observed.data();
                out.var_class = var_class.data();
                int rc = 0;
                if (n_labels) {
                    std::vector<uint64_t> label_off(n + 1, 0);
                    std::vector<uint32_t> label_idx;
                    rc = avf_strat_batch_labels(strat, genome, all, first + at, n, label_off.data(), nullptr);
                    if (!rc) {
                        label_idx.resize(label_off[n] + 1);
                        rc = avf_strat_batch_labels(strat, genome, all, first + at, n, label_off.data(), label_idx.data());
                    }
                    if (rc) {
                        worker_err[w] = std::string("cannot list the region labels: ") + avf_last_error();
                        break;
                    }
                    avk_dev_batch *db = nullptr;
                    uint64_t v_first = 0;
                    rc = upload_part(my, b, first + at, n, &db, &v_first);
                    if (!rc) rc = avk_compare_resident(my, db, &cfg, nullptr);
                    avk_result_batch shifted = out_for(out, v_first);
                    if (!rc) rc = avk_results_download(my, db, &shifted);
                    if (!rc) rc = avk_label_tallies(my, db, n_labels, label_off.data(), label_idx.data(), w_strat[w].data());
                    if (db) avk_batch_free(my, db);
                } else rc = compare_part(my, b, first + at, n, cfg, out);
                if (rc) {
                    worker_err[w] = std::string("compare failed: ") + avk_last_error(my);
                    break;
                }
                for (size_t k = 0; k < (size_t)AVK_TALLY_LEN; ++k) w_total[w][k] += w_tally[k];
                for (uint64_t r = 0; r < n; ++r)
                    if (out.status[r] != 0) {
                        std::lock_guard<std::mutex> lock(log_mutex);
                        report_unsolved(b, r, out.status[r]);
                    }
            }
            if (w > 0) avk_ctx_destroy(my);
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < n_workers; ++w) pool.emplace_back(worker, w);
        worker(0);
        for (std::thread &t : pool) t.join();
        for (size_t w = 0; w < n_workers; ++w)
            if (!worker_err[w].empty()) die(70, worker_err[w].c_str(), "");
        for (size_t w = 0; w < n_workers; ++w) {
            for (size_t k = 0; k < (size_t)AVK_TALLY_LEN; ++k) total[k] += w_total[w][k];
            for (size_t k = 0; k < strat_total.size(); ++k) strat_total[k] += w_strat[w][k];
        }
    }
    for (uint64_t at = 0; n_workers == 1 && at < count; at += batch_regions) {
        const uint64_t n = count - at < batch_regions ? count - at : batch_regions;
        avk_region_batch b = *all; /* a window of the region arrays; variant arrays are shared */
        b.n_regions = n;
        b.region_id = all->region_id + first + at;
        b.contig_idx = all->contig_idx + first + at;
        b.start = all->start + first + at;
        b.end = all->end + first + at;
        b.t_off = all->t_off + first + at;
        b.t_cnt = all->t_cnt + first + at;
        b.q_off = all->q_off + first + at;
        b.q_cnt = all->q_cnt + first + at;
        avk_result_batch out;
        memset(&out, 0, sizeof(out));
        out.status = status.data() + first + at;
        out.tally = tally.data();
        out.var_expected = var_expected.data(); /* indexed by the (shared) variant arrays */
        out.var_observed = var_observed.data();
        out.var_class = var_class.data();
        if (debug) {
            gm.resize((size_t)n * AVK_N_GROUPS * AVK_N_FIELDS);
            out.group_metrics = gm.data();
        }
        if (debug) {
            seq_off.resize(n);
            seq_stride.resize(n);
            seq_len.assign(5 * n, 0);
            uint64_t total_seq = 0;
            for (uint64_t r = 0; r < n; ++r) {
                seq_stride[r] = avk_seq_stride(&b, r);
                seq_off[r] = total_seq;
                total_seq += 5ull * seq_stride[r];
            }
            seq_bytes.resize(total_seq + 16);
            out.seq_bytes = seq_bytes.data();
            out.seq_off = seq_off.data();
            out.seq_stride = seq_stride.data();
            out.seq_len = seq_len.data();
        }
        if (device_labels) {
            std::vector<uint64_t> label_off(n + 1, 0);
            std::vector<uint32_t> label_idx;
            int rc_labels = 0;
            std::thread th_labels([&] { /* beside the upload and the kernels */
                rc_labels = avf_strat_batch_labels(strat, genome, all, first + at, n, label_off.data(), nullptr);
                if (!rc_labels) {
                    label_idx.resize(label_off[n] + 1);
                    rc_labels = avf_strat_batch_labels(strat, genome, all, first + at, n, label_off.data(), label_idx.data());
                }
            });
            avk_dev_batch *db = nullptr;
            uint64_t v_first = 0;
            int rc = upload_part(ctx, b, first + at, n, &db, &v_first);
            if (!rc) rc = avk_compare_resident(ctx, db, &cfg, nullptr);
            avk_result_batch shifted = out_for(out, v_first);
            if (!rc) rc = avk_results_download(ctx, db, &shifted);
            th_labels.join();
            if (rc) die(70, "compare failed", avk_last_error(ctx));
            if (rc_labels) die(70, "cannot list the region labels", avf_last_error());
            if (avk_label_tallies(ctx, db, n_labels, label_off.data(), label_idx.data(), strat_total.data())) die(70, "stratified tallies failed", avk_last_error(ctx));
            avk_batch_free(ctx, db);
        } else if (compare_part(ctx, b, first + at, n, cfg, out)) die(70, "compare failed", avk_last_error(ctx));
        for (size_t k = 0; k < (size_t)AVK_TALLY_LEN; ++k) total[k] += tally[k];
        if (debug && (avf_region_summary_rows(region_table, genome, all, first + at, n, out.status, gm.data()) ||
                      avf_region_sequences_rows(sequence_table, genome, all, first + at, n, out.status, seq_bytes.data(), seq_len.data(), seq_off.data(),
                                                seq_stride.data())))
            die(74, "Error while writing the debug tables", avf_last_error());
        for (uint64_t r = 0; n_labels && !device_labels && r < n; ++r) { /* debug runs: the blocks are on the host anyway */
            if (out.status[r] != 0) continue;
            const uint32_t hit = avf_strat_region_labels(strat, genome, all, first + at + r, labels.data(), n_labels);
            const uint32_t *block = gm.data() + (size_t)r * AVK_N_GROUPS * AVK_N_FIELDS;
            for (uint32_t h = 0; h < hit; ++h) {
                uint64_t *dst = strat_total.data() + (size_t)labels[h] * AVK_TALLY_LEN;
                for (int k = 0; k < AVK_N_GROUPS * AVK_N_FIELDS; ++k) dst[k] += block[k];
            }
        }
        for (uint64_t r = 0; r < n; ++r)
            if (out.status[r] != 0) report_unsolved(b, r, out.status[r]);
    }
    const double s_solve = seconds_since(t0);

    t0 = std::chrono::steady_clock::now();
    if (avf_table_close(region_table) || avf_table_close(sequence_table)) die(74, "Error while saving the debug tables", avf_last_error());
    const std::string summary = out_dir + "/summary.tsv";
    if (avf_write_summary_stratified(summary.c_str(), label.c_str(), total.data(), strat, strat_total.data(), mask))
        die(74, "Error while saving summary file", avf_last_error());
    /* the annotated VCFs (VariantCategorizer): truth.vcf.gz and query.vcf.gz with their .tbi */
    std::string command;
    for (int i = 0; i < argc; ++i) command += (i ? " " : "") + std::string(argv[i]);
    const char *in_vcf[2] = {truth.c_str(), query.c_str()}, *samples[2] = {truth_sample.c_str(), query_sample.c_str()}, *names[2] = {"/truth.vcf.gz", "/query.vcf.gz"};
    {
        int rcs[2] = {0, 0};
        std::string errs[2];
        auto write_one = [&](int src) {
            rcs[src] = avf_write_annotated_vcf((out_dir + names[src]).c_str(), in_vcf[src], samples[src], avk_version(), command.c_str(), genome, all, src,
                                               status.data(), var_expected.data(), var_observed.data(), var_class.data());
            if (rcs[src]) errs[src] = avf_last_error(); /* the error text is per thread */
        };
        std::thread other(write_one, 1);
        write_one(0);
        other.join();
        for (int src = 0; src < 2; ++src)
            if (rcs[src]) die(74, "Error while saving output files", errs[src].c_str());
    }
    const double s_write = seconds_since(t0);

    fprintf(stderr, "Solved:error blocks: %llu : %llu\n", (unsigned long long)total[AVK_TALLY_LEN - 2], (unsigned long long)total[AVK_TALLY_LEN - 1]);
    fprintf(stderr, "stages [s]: load %.3f (side by side: reference %.3f, truth calls %.3f, query calls %.3f, gpu context %.3f), regions %.3f (beside it: reference upload %.3f), "
                    "solve (pack + H2D + kernels + D2H) %.3f, summary + annotated VCFs %.3f\n",
            s_load, s_genome, s_calls[0], s_calls[1], s_ctx, s_feed, s_ref, s_solve, s_write);
    fprintf(stderr, "Comparisons completed in %.3f seconds (%.2f M regions/s in the solve stage).\n", seconds_since(t_start),
            s_solve > 0 ? (double)count / s_solve / 1e6 : 0.0);
    th_free.join();
    avk_ctx_destroy(ctx);
    avf_feed_free(feed);
    avf_strat_free(strat);
    avf_genome_free(genome);
    return 0;
}
