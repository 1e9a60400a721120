// cellector — host binary of the MI355X build: same command line (cellector/src/params.yml), same input files
// and the same output files as the reference's Rust binary (main.rs), with the scoring path running in
// libcellector_hip.so through the C ABI of include/cellector_ffi.h.  The reference host is Rust; this image has no
// Rust toolchain, so the host above the C ABI is C++ (INTEGRATION.md shows the equivalent Rust binding).
//
// Host-side pieces restated here (cited per function): load_params (main.rs:629-677), create_output_dir /
// load_barcodes / load_ground_truth / load_vcf_data (load_data.rs:37-107), the driver loop cellector()
// (main.rs:36-50), the writers output_iteration_tsv (main.rs:349-366), locus_filter_and_output_locus_data
// (main.rs:422-498), output_final_assignments + pretty_print (main.rs:133-226), output_final_vcf (main.rs:52-131).
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <optional>
#include <chrono>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../include/cellector_ffi.h"

namespace {

constexpr int EXIT_PANIC = 101;  // what a Rust panic gives the caller (cellector_pipeline.py checks != 0 only)
[[noreturn]] void die(int code, const std::string &msg)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(code);
}

// ---- Rust `{}` formatting of f64: shortest round-trip digits, never scientific --------
std::string fmt(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}
std::string fmt(uint64_t v) { return std::to_string(v); }
// the same, appended to a row under construction
void put(std::string &o, double v)
{
    if (std::isnan(v)) { o += "NaN"; return; }
    if (std::isinf(v)) { o += v > 0 ? "inf" : "-inf"; return; }
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    o.append(buf, r.ptr);
}
void put(std::string &o, uint64_t v)
{
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    o.append(buf, r.ptr);
}
// Formats rows [0, n) with row(i, out) on several host threads (contiguous ranges) and writes them in order: at 1M cells
// the reference-shaped fprintf loops were a visible part of the run once the scoring itself takes milliseconds.
template <class F>
void write_rows(FILE *f, uint64_t n, F row)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1 || n < 20000) nt = 1;
    std::vector<std::string> buf(nt);
    auto work = [&](unsigned t) {
        const uint64_t b = n * t / nt, e = n * (t + 1) / nt;
        buf[t].reserve((size_t)(e - b) * 96);
        for (uint64_t i = b; i < e; i++) row(i, buf[t]);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (unsigned t = 0; t < nt; t++) fwrite(buf[t].data(), 1, buf[t].size(), f);
}

// ---- reader (load_data.rs:240-251): ".gz" by extension, multi-member ------------------------------------------
struct Lines {
    gzFile gz = nullptr;
    std::string cur;
    explicit Lines(const std::string &path)
    {
        gz = gzopen(path.c_str(), "rb");  // transparent for plain files
        if (!gz) die(EXIT_PANIC, "couldn't open file " + path);
        gzbuffer(gz, 1 << 20);
    }
    ~Lines() { if (gz) gzclose(gz); }
    bool next(std::string &out)  // BufRead::lines(): strips "\n" and a preceding "\r"
    {
        out.clear();
        char buf[1 << 16];
        bool any = false;
        while (gzgets(gz, buf, sizeof buf)) {
            any = true;
            size_t n = strlen(buf);
            if (n && buf[n - 1] == '\n') {
                out.append(buf, n - 1);
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(buf, n);
        }
        return any;
    }
};

// the bytes of a (possibly gzipped) text file
std::string read_whole(const std::string &path)
{
    gzFile gz = gzopen(path.c_str(), "rb");  // transparent for plain files
    if (!gz) die(EXIT_PANIC, "couldn't open file " + path);
    gzbuffer(gz, 1 << 20);
    std::string out;
    std::vector<char> buf(4 << 20);
    for (;;) {
        const int n = gzread(gz, buf.data(), (unsigned)buf.size());
        if (n <= 0) break;
        out.append(buf.data(), (size_t)n);
    }
    gzclose(gz);
    return out;
}
// BufRead::lines() over a buffer: "\n" ends a line, a "\r" in front of it is dropped, a last line needs no terminator
std::vector<std::string_view> split_lines(const std::string &text)
{
    std::vector<std::string_view> out;
    out.reserve(text.size() / 16 + 1);
    size_t b = 0;
    while (b < text.size()) {
        size_t e = text.find('\n', b);
        const size_t next = e == std::string::npos ? text.size() : e + 1;
        if (e == std::string::npos) e = text.size();
        size_t len = e - b;
        if (len && e < text.size() && text[e] == '\n' && text[e - 1] == '\r') len--;  // (only a terminated line loses its "\r")
        out.emplace_back(text.data() + b, len);
        b = next;
    }
    return out;
}

std::vector<std::string> split(const std::string &s, char sep)
{
    std::vector<std::string> out;
    size_t b = 0;
    for (;;) {
        size_t e = s.find(sep, b);
        if (e == std::string::npos) { out.push_back(s.substr(b)); return out; }
        out.push_back(s.substr(b, e - b));
        b = e + 1;
    }
}

// ---- load_params (main.rs:629-677, params.yml) -----------------------------------------------------------------
struct Params {
    std::string ref_mtx, alt_mtx, barcodes, output_directory;
    std::optional<std::string> ground_truth, vcf;
    uint64_t min_alt = 4, min_ref = 4, min_alleles_posterior = 5, min_loci_used = 30;
    double posterior_threshold = 0.999, interquartile_range_multiple = 5.0;
    std::optional<double> expected_percent_minority;  // parsed, never used (quirk Q2)
    int device = -1;                                   // extension: ONE GPU to run on
    std::vector<int> devices;                          // extension: the GPUs to shard the cells over (default: GPU 0)
    bool devices_auto = false;                         // --devices auto: as many visible GPUs as the input can feed
};

const char *USAGE =
    "cellector 1.0.0\nHaynes Heaton <whheaton@gmail.com>\ngenotype outlier detection for scRNAseq\n\n"
    "USAGE:\n    cellector [OPTIONS] --alt <alt> --barcodes <barcodes> --output_directory <output_directory> --ref <ref>\n\n"
    "OPTIONS:\n"
    "    -a, --alt <alt>                                                    alt.mtx matrix from vartrix\n"
    "    -b, --barcodes <barcodes>                                          cell barcodes\n"
    "        --expected_percent_minority <expected_percent_minority>        percent of cells expected to come from the minority genotype\n"
    "    -g, --ground_truth <ground_truth>                                  cell hashing assignments or other ground truth\n"
    "        --interquartile_range_multiple <interquartile_range_multiple>  IQR multiples below the 25th percentile for the outlier threshold\n"
    "        --min_alleles_posterior <min_alleles_posterior>                minimum alleles per distribution for the posterior calculation\n"
    "        --min_alt <min_alt>                                            minimum number of cells containing the alt allele (default 4)\n"
    "        --min_loci_for_assignment <min_loci_for_assignment>            minimum loci to assign a cell (default 30)\n"
    "        --min_ref <min_ref>                                            minimum number of cells containing the ref allele (default 4)\n"
    "        --output_directory <output_directory>                          output directory\n"
    "        --posterior_threshold <posterior_threshold>                    posterior threshold for assignment (default 0.999)\n"
    "    -r, --ref <ref>                                                    ref.mtx matrix from vartrix\n"
    "    -v, --vcf <vcf>                                                    vcf associated with alt.mtx and ref.mtx\n"
    "        --device <n>                                                   run on this one GPU (not in the reference)\n"
    "        --devices <a,b,...>                                            GPUs to shard the cells over, RCCL exchanges between them (not in\n"
    "                                                                       the reference; default: GPU 0; `auto`: one visible GPU per 4 GB of\n"
    "                                                                       alt.mtx text; a GPU listed twice = two logical shards on it)\n";

uint64_t parse_usize(const std::string &name, const std::string &s)
{
    uint64_t v = 0;
    auto r = std::from_chars(s.data(), s.data() + s.size(), v);
    if (s.empty() || r.ec != std::errc() || r.ptr != s.data() + s.size())
        die(EXIT_PANIC, "invalid value '" + s + "' for --" + name + ": expected an unsigned integer");
    return v;
}
double parse_f64(const std::string &name, const std::string &s)
{
    char *end = nullptr;
    double v = strtod(s.c_str(), &end);
    if (s.empty() || end != s.c_str() + s.size()) die(EXIT_PANIC, "invalid value '" + s + "' for --" + name + ": expected a number");
    return v;
}

Params load_params(int argc, char **argv)
{
    static const std::map<std::string, std::string> shorts = {{"-r", "ref"}, {"-a", "alt"}, {"-b", "barcodes"},
                                                              {"-g", "ground_truth"}, {"-v", "vcf"}};
    static const char *known[] = {"output_directory", "ref", "alt", "barcodes", "min_alt", "min_ref", "ground_truth",
                                  "vcf", "posterior_threshold", "interquartile_range_multiple", "min_alleles_posterior",
                                  "expected_percent_minority", "min_loci_for_assignment", "device", "devices"};
    std::map<std::string, std::string> got;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], name, value;
        bool have_value = false;
        if (a == "-h" || a == "--help") { fputs(USAGE, stdout); exit(0); }
        if (a == "-V" || a == "--version") { puts("cellector 1.0.0"); exit(0); }
        if (a.rfind("--", 0) == 0) {
            size_t eq = a.find('=');
            name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { value = a.substr(eq + 1); have_value = true; }
        } else if (shorts.count(a.substr(0, 2))) {
            name = shorts.at(a.substr(0, 2));
            if (a.size() > 2) { value = a.substr(a[2] == '=' ? 3 : 2); have_value = true; }
        } else {
            die(1, "error: Found argument '" + a + "' which wasn't expected, or isn't valid in this context\n\n" + USAGE);
        }
        if (std::find_if(std::begin(known), std::end(known), [&](const char *k) { return name == k; }) == std::end(known))
            die(1, "error: Found argument '" + a + "' which wasn't expected, or isn't valid in this context\n\n" + USAGE);
        if (!have_value) {
            if (i + 1 >= argc) die(1, "error: The argument '--" + name + " <" + name + ">' requires a value but none was supplied");
            value = argv[++i];
        }
        if (got.count(name)) die(1, "error: The argument '--" + name + " <" + name + ">' was provided more than once");
        got[name] = value;
    }
    for (const char *req : {"alt", "barcodes", "output_directory", "ref"})
        if (!got.count(req)) die(1, std::string("error: The following required arguments were not provided:\n    --") + req + " <" + req + ">\n\n" + USAGE);
    Params p;
    p.ref_mtx = got["ref"]; p.alt_mtx = got["alt"]; p.barcodes = got["barcodes"]; p.output_directory = got["output_directory"];
    if (got.count("ground_truth")) p.ground_truth = got["ground_truth"];
    if (got.count("vcf")) p.vcf = got["vcf"];
    if (got.count("min_alt")) p.min_alt = parse_usize("min_alt", got["min_alt"]);
    if (got.count("min_ref")) p.min_ref = parse_usize("min_ref", got["min_ref"]);
    if (got.count("posterior_threshold")) p.posterior_threshold = parse_f64("posterior_threshold", got["posterior_threshold"]);
    if (got.count("interquartile_range_multiple"))
        p.interquartile_range_multiple = parse_f64("interquartile_range_multiple", got["interquartile_range_multiple"]);
    if (got.count("min_alleles_posterior")) p.min_alleles_posterior = parse_usize("min_alleles_posterior", got["min_alleles_posterior"]);
    if (got.count("expected_percent_minority")) p.expected_percent_minority = parse_f64("expected_percent_minority", got["expected_percent_minority"]);
    if (got.count("min_loci_for_assignment")) p.min_loci_used = parse_usize("min_loci_for_assignment", got["min_loci_for_assignment"]);
    if (got.count("device")) p.device = (int)parse_usize("device", got["device"]);
    if (got.count("devices")) {
        if (got["devices"] == "auto") p.devices_auto = true;
        else
            for (const std::string &t : split(got["devices"], ',')) p.devices.push_back((int)parse_usize("devices", t));
    }
    if (got.count("device") && got.count("devices")) die(1, "error: The argument '--device <n>' cannot be used with '--devices <a,b,...>'");
    return p;
}

// ---- statrs pieces for the VCF genotype rule -----------------------------------
double ln_gamma(double x)
{
    static const double dk[11] = {2.48574089138753565546e-5,  1.05142378581721974210,    -3.45687097222016235469,
                                  4.51227709466894823700,     -2.98285225323576655721,   1.05639711577126713077,
                                  -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                                  4.63399473359905636708e-6,  -2.71994908488607703910e-9};
    double s = dk[0];
    for (int i = 1; i <= 10; i++) s += dk[i] / (x + (double)i - 1.0);
    return std::log(s) + 0.6207822376352452223455184457816472122518527279025978 +
           (x - 0.5) * std::log((x - 0.5 + 10.900511) / 2.71828182845904523536028747135266250);
}
double ln_factorial(uint64_t x)
{
    struct Cache {  // statrs FCACHE: running f64 product (built once; thread-safe initialisation)
        double v[171];
        Cache()
        {
            v[0] = 1.0;
            for (int i = 1; i <= 170; i++) v[i] = v[i - 1] * (double)i;
        }
    };
    static const Cache cache;
    return x <= 170 ? std::log(cache.v[x]) : ln_gamma((double)x + 1.0);
}
double binomial_pmf(double p, uint64_t n, uint64_t k)
{
    if (k > n) return 0.0;
    if (p == 0.0) return k == 0 ? 1.0 : 0.0;
    if (std::fabs(p - 1.0) <= 4 * 2.220446049250313e-16) return k == n ? 1.0 : 0.0;
    const double lnb = ln_factorial(n) - ln_factorial(k) - ln_factorial(n - k);
    return std::exp(lnb + (double)k * std::log(p) + (double)(n - k) * std::log(1.0 - p));
}
// statrs Data::median on a copy (main.rs:442-443); NaN for empty data
double median_of(std::vector<double> v)
{
    if (v.empty()) return NAN;
    std::sort(v.begin(), v.end());
    const size_t k = v.size() / 2;
    return v.size() % 2 ? v[k] : (v[k - 1] + v[k]) / 2.0;
}

struct Ctx {
    cellector_ctx *c = nullptr;
    void ck(cellector_status s, const char *what)
    {
        if (s != CELLECTOR_OK) die(EXIT_PANIC, std::string(what) + ": " + (c ? cellector_last_error(c) : "no context"));
    }
};

FILE *create(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) die(EXIT_PANIC, "Unable to create file " + path);
    static std::vector<char> *bufs = new std::vector<char>[64];
    static int nb = 0;
    if (nb < 64) { bufs[nb].resize(1 << 20); setvbuf(f, bufs[nb].data(), _IOFBF, bufs[nb].size()); nb++; }
    return f;
}

struct VcfLocus { std::string chrom, pos; };

}  // namespace

int main(int argc, char **argv)
{
    const Params params = load_params(argc, argv);
    // create_output_dir (load_data.rs:66-71): non-recursive mkdir, failure ignored (quirk Q13)
    (void)mkdir(params.output_directory.c_str(), 0777);

    // load_barcodes (load_data.rs:73-83): the lines of the file; barcode -> cell index for the ground truth, a later
    // duplicate overwriting an earlier one (HashMap::insert).  The file is read whole and the table is a flat open-addressing
    // one over views into it — a std::unordered_map<std::string, size_t> of a million barcodes was 0.3-0.6 s of node
    // allocations.
    std::string barcode_text = read_whole(params.barcodes);
    std::vector<std::string_view> barcodes = split_lines(barcode_text);
    std::vector<uint32_t> bc_slot;  // line index + 1 of the LAST line with that barcode, 0 = empty
    size_t n_distinct = 0;
    {
        size_t cap = 16;
        while (cap < 2 * barcodes.size() + 2) cap <<= 1;
        bc_slot.assign(cap, 0u);
        if (barcodes.size() >= 0xffffffffull) die(EXIT_PANIC, "too many barcodes");
        for (size_t i = 0; i < barcodes.size(); i++) {
            size_t h = std::hash<std::string_view>{}(barcodes[i]) & (cap - 1);
            for (;; h = (h + 1) & (cap - 1)) {
                if (!bc_slot[h]) { bc_slot[h] = (uint32_t)i + 1; n_distinct++; break; }
                if (barcodes[bc_slot[h] - 1] == barcodes[i]) { bc_slot[h] = (uint32_t)i + 1; break; }
            }
        }
    }
    auto barcode_to_cell = [&](std::string_view key) -> size_t {  // SIZE_MAX: not a barcode
        const size_t cap = bc_slot.size();
        for (size_t h = std::hash<std::string_view>{}(key) & (cap - 1);; h = (h + 1) & (cap - 1)) {
            if (!bc_slot[h]) return SIZE_MAX;
            if (barcodes[bc_slot[h] - 1] == key) return bc_slot[h] - 1;
        }
    };
    // load_ground_truth (load_data.rs:85-107): one label per DISTINCT barcode (the vector has the map's length)
    std::vector<std::string> ground_truth(n_distinct, "na");
    if (params.ground_truth) {
        Lines in(*params.ground_truth);
        std::string line;
        while (in.next(line)) {
            auto cols = split(line, '\t');
            if (cols.size() != 2) die(EXIT_PANIC, "Invalid line format: " + line + "\nThe correct format is: barcode\tassignment");
            const size_t cell = barcode_to_cell(cols[0]);
            if (cell != SIZE_MAX && cell < ground_truth.size()) ground_truth[cell] = cols[1];
        }
    }

    // CELLECTOR_TIMING=1: phase wall times on stderr (not part of the reference's output)
    const bool timing = getenv("CELLECTOR_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "[timing] %-24s %8.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    // load_cell_data (load_data.rs:134-181) on the device
    // Which GPUs: --device n / --devices a,b,... as given; neither (cellector_pipeline.py:223-226 passes neither flag): GPU 0 —
    // the multi-GPU path over RCCL is opt-in until it has run on a multi-GPU node (README).  --devices auto: all visible GPUs
    // when the input is large enough to feed them (one GPU per 4 GB of alt.mtx text, i.e. >= 1.3e8 entries each: below that
    // the per-iteration exchanges and the communicator set-up cost more than the extra GPUs save); should the multi-device
    // ctx fail to come up there, the run falls back to GPU 0 with a note on stderr.
    Ctx g;
    std::vector<int> devices = params.devices;
    if (devices.empty() && params.device >= 0) devices.push_back(params.device);
    if (devices.empty() && !params.devices_auto) devices.push_back(0);
    if (devices.empty()) {
        int visible = 0;
        (void)cellector_device_count(&visible);
        struct stat sb;
        const uint64_t alt_bytes = stat(params.alt_mtx.c_str(), &sb) == 0 ? (uint64_t)sb.st_size : 0;
        const bool gz = params.alt_mtx.size() > 3 && params.alt_mtx.compare(params.alt_mtx.size() - 3, 3, ".gz") == 0;
        const uint64_t text_bytes = alt_bytes * (gz ? 4 : 1);  // (text is ~4x the .gz)
        const uint64_t want = text_bytes < (4ull << 30) ? 1 : (text_bytes + (4ull << 30) - 1) / (4ull << 30);  // (BASELINE cfg5: 30.7 GB -> 8)
        if (const char *e = getenv("CELLECTOR_DEVICES_AUTO_MAX")) visible = std::min(visible, atoi(e));
        const int n = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)std::min(visible, 16)));
        for (int i = 0; i < n; i++) devices.push_back(i);
    }
    cellector_status cst = cellector_create_multi(&g.c, devices.data(), (int)devices.size());
    if (cst != CELLECTOR_OK && params.devices_auto && devices.size() > 1) {
        fprintf(stderr, "cellector: the %zu-GPU context did not come up (status %d); running on GPU 0\n", devices.size(), (int)cst);
        devices.assign(1, 0);
        cst = cellector_create_multi(&g.c, devices.data(), 1);
    }
    if (cst != CELLECTOR_OK)
        die(EXIT_PANIC, "cellector: no usable MI355X device (asked for " + std::to_string(devices.size()) + ", first: " +
                            std::to_string(devices[0]) + "; there is no CPU fallback)");
    if (const char *e = getenv("CELLECTOR_ENGINE")) g.ck(cellector_set_option(g.c, "engine", atoi(e)), "engine");
    if (const char *e = getenv("CELLECTOR_BANK_ORDER")) g.ck(cellector_set_option(g.c, "bank_order", atoi(e)), "bank_order");
    g.ck(cellector_set_option(g.c, "keep_coo", params.vcf ? 1 : 0), "option");
    lap("barcodes + device init");
    g.ck(cellector_load_mtx(g.c, params.alt_mtx.c_str(), params.ref_mtx.c_str(), params.min_alt, params.min_ref), "load_cell_data");
    lap("load_mtx (text -> device)");
    cellector_dims_t dm;
    g.ck(cellector_dims(g.c, &dm), "dims");
    const uint64_t N = dm.total_cells, L = dm.loci_used;
    if (barcodes.size() < N || ground_truth.size() < N)  // init_cell_data indexes both (load_data.rs:231-232)
        die(EXIT_PANIC, "index out of bounds: the barcodes file has " + std::to_string(barcodes.size()) +
                            " lines but the matrix has " + std::to_string(N) + " cells");
    std::vector<uint64_t> locus_ids(L);
    std::vector<uint32_t> entries_per_cell(N);
    g.ck(cellector_locus_ids(g.c, locus_ids.data()), "locus_ids");
    g.ck(cellector_entries_per_cell(g.c, entries_per_cell.data()), "entries_per_cell");

    // load_vcf_data (load_data.rs:37-63)
    std::vector<VcfLocus> vcf_data;
    if (params.vcf) {
        Lines in(*params.vcf);
        std::string line;
        while (in.next(line)) {
            if (!line.empty() && line[0] == '#') continue;
            auto t = split(line, '\t');
            if (t.size() < 5) die(EXIT_PANIC, "index out of bounds: vcf record with fewer than 5 columns: " + line);
            vcf_data.push_back({t[0], t[1]});
        }
    }

    // cellector() (main.rs:36-50)
    std::vector<double> ll(N), ell(N), nloci(N), norm(N);
    std::vector<double> c_min(L), c_maj(L);
    std::vector<uint64_t> n_min(L), n_maj(L), a_min(L), r_min(L), a_maj(L), r_maj(L);
    const std::string &od = params.output_directory;
    for (uint64_t iteration = 0;; iteration++) {
        cellector_iter_summary s;
        g.ck(cellector_em_iteration(g.c, params.interquartile_range_multiple, &s), "compute_new_excluded");
        printf("detected %llu new anomylous cells and rescued %llu cells to the majority in iteration %llu\n",
               (unsigned long long)s.n_new_excluded, (unsigned long long)s.n_rescued, (unsigned long long)(iteration + 1));
        printf("median normalized log likelihood %s with interquartile range %s, threshold %s\n", fmt(s.median).c_str(),
               fmt(s.iqr).c_str(), fmt(s.threshold).c_str());
        if (s.n_near_threshold)  // stderr only: stdout stays byte-compatible with main.rs:338-339
            fprintf(stderr, "warning: iteration %llu: %llu cell(s) within 1e-9 (relative) of the threshold %s; the device's "
                            "log-pmf arithmetic differs from the reference's by ~1e-11, so their anomaly flag may differ from "
                            "the reference's\n", (unsigned long long)(iteration + 1), (unsigned long long)s.n_near_threshold,
                    fmt(s.threshold).c_str());
        g.ck(cellector_iter_cell_outputs(g.c, ll.data(), ell.data(), nloci.data(), norm.data()), "cell outputs");
        g.ck(cellector_iter_locus_outputs(g.c, c_min.data(), c_maj.data(), n_min.data(), n_maj.data(), a_min.data(),
                                          r_min.data(), a_maj.data(), r_maj.data()), "locus outputs");
        {   // locus_filter_and_output_locus_data (main.rs:422-498)
            FILE *f = create(od + "/iteration_" + std::to_string(iteration) + "_locus_contribution.tsv");
            fputs("locus_id\tchrom\tpos\tlog_likelihood_minority\tlog_likelihood_majority\texpected_loglike_minority\t"
                  "expected_loglike_majority\tminority_cellcount\tmajority_cellcount\tlog_likelihood_minority_per_cell\t"
                  "log_likelihood_majority_per_cell\tminority_alt\tminority_ref\tmajority_alt\tmajority_ref\tminority_af\t"
                  "majority_af\n", f);
            std::vector<double> pc_min(L), pc_maj(L), for_thr;
            for (uint64_t l = 0; l < L; l++) {
                if (n_min[l]) { pc_min[l] = c_min[l] / (double)n_min[l]; for_thr.push_back(pc_min[l]); } else pc_min[l] = 0.0;
                pc_maj[l] = n_maj[l] ? c_maj[l] / (double)n_maj[l] : 0.0;
            }
            std::vector<size_t> order(L);
            std::iota(order.begin(), order.end(), (size_t)0);
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return pc_min[a] < pc_min[b]; });
            const double med = median_of(for_thr);
            for (uint64_t l = 0; l < L; l++)
                if (pc_min[l] < -80.0)  // main.rs:444-449; the mask itself was updated on the device
                    printf("filtering locus %llu locus index %llu because it was contributing %s vs median %s per cell to "
                           "log likelihood of minority cells\n", (unsigned long long)locus_ids[l], (unsigned long long)l,
                           fmt(pc_min[l]).c_str(), fmt(med).c_str());
            if (params.vcf)
                for (uint64_t l = 0; l < L; l++)
                    if (locus_ids[l] >= vcf_data.size()) die(EXIT_PANIC, "index out of bounds: vcf has fewer records than loci");
            write_rows(f, L, [&](uint64_t i, std::string &o) {
                const size_t l = order[i];
                const double af_min = a_min[l] + r_min[l] ? (double)a_min[l] / (double)(a_min[l] + r_min[l]) : 0.0;
                const double af_maj = a_maj[l] + r_maj[l] ? (double)a_maj[l] / (double)(a_maj[l] + r_maj[l]) : 0.0;
                put(o, locus_ids[l]); o += '\t';
                if (params.vcf) { o += vcf_data[locus_ids[l]].chrom; o += '\t'; o += vcf_data[locus_ids[l]].pos; }
                else o += "na\tna";
                o += '\t'; put(o, c_min[l]); o += '\t'; put(o, c_maj[l]);
                o += '\t'; put(o, c_min[l]); o += '\t'; put(o, c_maj[l]);  // quirk Q6: "expected" == plain contribution
                o += '\t'; put(o, n_min[l]); o += '\t'; put(o, n_maj[l]);
                o += '\t'; put(o, pc_min[l]); o += '\t'; put(o, pc_maj[l]);
                o += '\t'; put(o, a_min[l]); o += '\t'; put(o, r_min[l]); o += '\t'; put(o, a_maj[l]); o += '\t'; put(o, r_maj[l]);
                o += '\t'; put(o, af_min); o += '\t'; put(o, af_maj); o += '\n';
            });
            fclose(f);
        }
        {   // output_iteration_tsv (main.rs:349-366)
            FILE *f = create(od + "/iteration_" + std::to_string(iteration) + ".tsv");
            fputs("cell_id\tbarcode\tassignment\tlog_likelihood\texpected_log_likelihood\tnum_loci_used\n", f);
            write_rows(f, N, [&](uint64_t c, std::string &o) {
                put(o, c); o += '\t'; o += barcodes[c]; o += '\t'; o += ground_truth[c];
                o += '\t'; put(o, ll[c]); o += '\t'; put(o, ell[c]); o += '\t'; put(o, nloci[c]); o += '\n';
            });
            fclose(f);
            f = create(od + "/iteration_" + std::to_string(iteration) + "_threshold.tsv");
            fputs(fmt(s.threshold).c_str(), f);
            fclose(f);
        }
        if (!s.any_change) break;
    }

    lap("EM loop + iteration files");
    // calculate_posteriors (main.rs:228-280)
    std::vector<double> posterior(N), doublet(N), ll_maj(N), ll_min(N);
    std::vector<uint8_t> excluded(N);
    g.ck(cellector_posteriors(g.c, posterior.data(), doublet.data(), ll_maj.data(), ll_min.data()), "calculate_posteriors");
    g.ck(cellector_excluded(g.c, excluded.data()), "excluded");
    lap("posteriors");

    // output_final_vcf (main.rs:52-131)
    if (params.vcf) {
        const uint64_t TL = dm.total_loci;
        std::vector<uint64_t> amin(TL), rmin(TL), amaj(TL), rmaj(TL);
        g.ck(cellector_final_allele_tallies(g.c, amin.data(), rmin.data(), amaj.data(), rmaj.data()), "load_mtx_final");
        Lines in(*params.vcf);
        FILE *f = create(od + "/cellector.vcf");
        std::string line;
        uint64_t rec = 0;
        const double ambient = 0.03, gt_thr = 0.99;
        // the lines first (kind: 0 = "##" line, 1 = "#CHROM" line, 2 = record with its index), then formatted in parallel
        std::vector<std::string> lines;
        std::vector<uint64_t> rec_of;
        while (in.next(line)) {
            uint64_t kind = ~0ull;
            if (line.rfind("##", 0) == 0) kind = ~0ull;
            else if (line.rfind("#CHROM", 0) == 0) kind = ~0ull - 1;
            else {
                if (rec >= TL) die(EXIT_PANIC, "index out of bounds: vcf has more records than the matrix has loci");
                kind = rec++;
            }
            lines.push_back(line);
            rec_of.push_back(kind);
        }
        write_rows(f, lines.size(), [&](uint64_t i, std::string &o) {
            const std::string &ln = lines[i];
            if (rec_of[i] == ~0ull) { o += ln; o += '\n'; return; }
            if (rec_of[i] == ~0ull - 1) { o += ln; o += "\tmajority\tminority\n"; return; }
            const uint64_t r_ = rec_of[i];
            const uint64_t tot_alt = amin[r_] + amaj[r_], tot_ref = rmin[r_] + rmaj[r_];
            const double soup = tot_alt + tot_ref > 0 ? (double)tot_alt / (double)(tot_alt + tot_ref) : 0.5;
            const double p_alt = (1.0 - ambient) * 0.99 + ambient * soup, p_het = (1.0 - ambient) * 0.5 + ambient * soup,
                         p_ref = (1.0 - ambient) * 0.01 + ambient * soup;
            const char *gt[2];
            double mx[2];
            for (int w = 0; w < 2; w++) {  // 0 = majority, 1 = minority
                const uint64_t a = w ? amin[r_] : amaj[r_], r = w ? rmin[r_] : rmaj[r_];
                const double l_alt = binomial_pmf(p_alt, a + r, a), l_het = binomial_pmf(p_het, a + r, a),
                             l_ref = binomial_pmf(p_ref, a + r, a);
                const double den = 1.0 / 3.0 * l_alt + 1.0 / 3.0 * l_het + 1.0 / 3.0 * l_ref;
                const double q_alt = l_alt * 1.0 / 3.0 / den, q_het = l_het * 1.0 / 3.0 / den, q_ref = l_ref * 1.0 / 3.0 / den;
                mx[w] = std::fmax(std::fmax(q_alt, q_het), q_ref);
                gt[w] = q_alt > gt_thr ? "1/1" : q_het > gt_thr ? "0/1" : q_ref > gt_thr ? "0/0" : "./.";
            }
            o += ln; o += "\tGT:GP:AO:RO\t";
            o += gt[0]; o += ':'; put(o, mx[0]); o += ':'; put(o, amaj[r_]); o += ':'; put(o, rmaj[r_]); o += '\t';
            o += gt[1]; o += ':'; put(o, mx[1]); o += ':'; put(o, amin[r_]); o += ':'; put(o, rmin[r_]); o += '\n';
        });
        fclose(f);
    }

    lap("final tallies + cellector.vcf");
    // output_final_assignments (main.rs:133-174)
    std::map<std::string, std::map<std::string, uint64_t>> assignment_gt_counts;
    std::map<std::string, uint64_t> gt_counts;
    {
        FILE *f = create(od + "/cellector_assignments.tsv");
        fputs("barcode\tposterior_assignment\tanomally_assignment\tlog_likelihood_loci_normalized\tloci_used\t"
              "posterior_assign_qual\tmajority_log_likelihood\tminority_log_likelihood\tground_truth_assignment\n", f);
        static const char *const PA[4] = {"unassigned", "0", "1", "doublet"};
        std::vector<uint8_t> pa_of(N);
        for (uint64_t c = 0; c < N; c++) {
            int pa = 0;
            if (posterior[c] > params.posterior_threshold) pa = 1;
            else if (1.0 - posterior[c] > params.posterior_threshold) pa = 2;
            if (doublet[c] > 0.5) pa = 3;
            if (entries_per_cell[c] < params.min_loci_used) pa = 0;  // quirk Q5
            pa_of[c] = (uint8_t)pa;
            assignment_gt_counts[PA[pa]][ground_truth[c]]++;
            gt_counts[ground_truth[c]]++;
        }
        write_rows(f, N, [&](uint64_t c, std::string &o) {
            const double post = std::fmax(posterior[c], 1.0 - posterior[c]);
            double q = std::fmin(-10.0 * std::log10(1.0 - post), 255.0);  // f64::min ignores NaN
            const uint64_t qual = (q != q || q < 0.0) ? 0 : (uint64_t)q;    // `as usize` saturates
            o += barcodes[c]; o += '\t'; o += PA[pa_of[c]]; o += '\t'; o += excluded[c] ? "0" : "1";
            o += '\t'; put(o, norm[c]); o += '\t'; put(o, (uint64_t)nloci[c]); o += '\t'; put(o, qual);
            o += '\t'; put(o, ll_maj[c]); o += '\t'; put(o, ll_min[c]); o += '\t'; o += ground_truth[c]; o += '\n';
        });
        fclose(f);
    }
    lap("assignments file");
    {   // pretty_print (main.rs:177-226); ties in the count sort are in hash order there, by name here
        std::vector<std::pair<std::string, uint64_t>> cv(gt_counts.begin(), gt_counts.end());
        std::stable_sort(cv.begin(), cv.end(), [](const auto &a, const auto &b) { return a.second > b.second; });
        const std::string first_header = "cellector assignment   ", header = "      0      1      unassigned\n";
        std::string sb = first_header + header;
        size_t xoffset = std::max<size_t>(3, first_header.size() + 2);
        sb += "cell_hashing";
        sb += std::string(xoffset >= 12 ? xoffset - 12 : 0, ' ') + "|" + std::string(header.size() - 1, '-') + "|\n";
        auto get = [&](const char *k, const std::string &gt) -> uint64_t {
            auto it = assignment_gt_counts.find(k);
            if (it == assignment_gt_counts.end()) return 0;
            auto jt = it->second.find(gt);
            return jt == it->second.end() ? 0 : jt->second;
        };
        for (auto &[gt, cnt] : cv) {
            (void)cnt;
            xoffset = std::max(xoffset, gt.size() + 3);
            const std::string c0 = fmt(get("0", gt)), c1 = fmt(get("1", gt)), un = fmt(get("unassigned", gt));
            sb += gt;
            const size_t glen = gt.size() ? gt.size() - 1 : 0;
            sb += std::string(xoffset >= glen ? xoffset - glen : 0, ' ');
            sb += " |  " + c0 + std::string(c0.size() < 4 ? 4 - c0.size() : 0, ' ');
            sb += " |  " + c1 + std::string(c1.size() < 4 ? 4 - c1.size() : 0, ' ');
            sb += " |  " + un + std::string(un.size() < 12 ? 12 - un.size() : 0, ' ') + "|\n";
        }
        sb += std::string(xoffset, ' ') + "|" + std::string(header.size() - 1, '-') + "|\n";
        printf("\n\n%s\n", sb.c_str());
    }
    // Every output file is closed: leave without tearing the context down.  Handing ~150 GB of device memory back block
    // by block (and destroying the host vectors) was half a second of the 1M x 200k run; the driver reclaims a dead
    // process' memory in one go.
    fflush(stdout);
    fflush(stderr);
    // (a multi-device ctx always leaves the orderly way: its communicator and worker threads are torn down in order;
    //  so does a run under a profiler or leak checker, CELLECTOR_TEARDOWN=1)
    if (devices.size() > 1 || getenv("CELLECTOR_TEARDOWN")) {
        cellector_destroy(g.c);
        return 0;
    }
    _exit(0);
}
