// combiner — synthetic-mixture maker: mixes two vartrix datasets into one majority / minority
// experiment with ground truth, the way the reference's `combiner` does (combiner/src/main.rs:23-50), so that labelled
// benchmarks of the scoring path can be made from real or synthetic inputs.  Same command line (combiner/src/params.yml),
// same output files:
//
//   get_locus_mapping (main.rs:197-231)   loci of dataset 2 are renumbered into dataset 1's numbering by (chrom, pos) of the two
//                                         VCFs; positions dataset 1 does not have are appended after its last locus
//   select_cells (main.rs:246-255)        a seeded sample of num_cells_1 / num_cells_2 cells, or the cells of --dataset2_mask
//   get_cell_ids_and_output_barcodes      output cell ids 1..n1 (dataset 1, "majority") then n1+1.. (dataset 2, "minority");
//   (main.rs:141-188)                     barcodes.tsv, gt.tsv; a dataset-2 barcode has its last character replaced by '2'
//   output_new_mtxs (main.rs:52-116)      per read an independent removal with probability --downsample_rate; entries of the
//                                         selected cells (kept even when both counts reach zero), sorted by (locus, cell, ref,
//                                         alt); alt.mtx / ref.mtx with tab-separated `locus cell count` lines, header entry
//                                         count 0 (which is why cellector never trusts that field)
//   stdout                                "n1,n2"
//
// What is NOT the reference's: the random stream.  The reference draws from rand 0.7's StdRng; this tool uses splitmix64
// (documented below), so for one seed the two programs pick different cells and drop different reads — the files have the same
// format and the same statistics.  No fixture of the reference depends on its stream (it ships none).
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <optional>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

[[noreturn]] void die(int code, const std::string &msg)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(code);
}
constexpr int EXIT_PANIC = 101;  // what a Rust panic gives the caller

// reader (main.rs:233-244): ".gz" by extension, multi-member; BufRead::lines() strips "\n" and a preceding "\r"
struct Lines {
    gzFile gz = nullptr;
    explicit Lines(const std::string &path)
    {
        gz = gzopen(path.c_str(), "rb");
        if (!gz) die(EXIT_PANIC, "couldn't open file " + path);
        gzbuffer(gz, 1 << 20);
    }
    ~Lines() { if (gz) gzclose(gz); }
    bool next(std::string &out)
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

uint64_t parse_usize(const std::string &what, const std::string &s)
{
    uint64_t v = 0;
    const char *b = s.data(), *e = b + s.size();
    if (b < e && *b == '+') b++;
    auto r = std::from_chars(b, e, v);
    if (b == e || r.ec != std::errc() || r.ptr != e) die(EXIT_PANIC, "cannot parse '" + s + "' as an unsigned integer (" + what + ")");
    return v;
}

// split_whitespace
std::vector<std::string> tokens(const std::string &s)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

// the tool's random stream: splitmix64 on a counter started at seed * golden ratio; uniform f64 in [0, 1) from the top 53 bits
struct Rng {
    uint64_t state;
    explicit Rng(uint64_t seed) : state(seed * 0x9E3779B97F4A7C15ull + 0x5EEDull) {}
    uint64_t next()
    {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return (uint64_t)(uniform() * (double)n) % n; }  // (n < 2^53)
};

struct Params {
    std::string vcf1, vcf2, alt1, ref1, alt2, ref2, barcodes1, barcodes2, output_directory;
    uint64_t num_cells_1 = 0;
    std::optional<uint64_t> num_cells_2;
    std::optional<std::string> dataset2_mask;
    uint64_t seed = 4;  // "guaranteed random number by fair dice roll" (main.rs:337)
    double downsample_rate = 0.0;
};

const char *USAGE =
    "combiner 1.0.0\ncombines data from genotype scRNAseq experiments (output from vartix)\n\n"
    "USAGE:\n    combiner [OPTIONS] --alt1 <alt1> --alt2 <alt2> --barcodes1 <barcodes1> --barcodes2 <barcodes2> --num_cells_1 <num_cells_1> "
    "--output_directory <output_directory> --ref1 <ref1> --ref2 <ref2> --vcf1 <vcf1> --vcf2 <vcf2>\n\n"
    "OPTIONS:\n"
    "        --alt1 <alt1>                          alt.mtx matrix from vartrix for dataset1\n"
    "        --alt2 <alt2>                          alt.mtx matrix from vartrix for dataset2\n"
    "        --barcodes1 <barcodes1>                cell barcodes for dataset1\n"
    "        --barcodes2 <barcodes2>                cell barcodes for dataset2\n"
    "        --dataset2_mask <dataset2_mask>        barcodes of cells include from dataset2 if not using the argument --num_cells_2\n"
    "        --downsample_rate <downsample_rate>    downsample data (probability not percent so 0.2 not 20) default 0.0\n"
    "        --num_cells_1 <num_cells_1>            number of cells to use from dataset1\n"
    "        --num_cells_2 <num_cells_2>            number of cells to use from dataset2\n"
    "    -o, --output_directory <output_directory>  name of output directory to put files\n"
    "        --ref1 <ref1>                          ref.mtx matrix from vartrix for dataset1\n"
    "        --ref2 <ref2>                          ref.mtx matrix from vartrix for dataset2\n"
    "        --seed <seed>                          set random number generator seed for deterministic behavior\n"
    "        --vcf1 <vcf1>                          variant file for dataset 1\n"
    "        --vcf2 <vcf2>                          variant file for dataset 2\n";

Params load_params(int argc, char **argv)
{
    static const char *known[] = {"output_directory", "vcf1", "vcf2", "ref1", "alt1", "ref2", "alt2", "barcodes1", "barcodes2",
                                  "num_cells_1", "num_cells_2", "dataset2_mask", "seed", "downsample_rate"};
    std::map<std::string, std::string> got;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], name, value;
        bool have = false;
        if (a == "-h" || a == "--help") { fputs(USAGE, stdout); exit(0); }
        if (a == "-V" || a == "--version") { puts("combiner 1.0.0"); exit(0); }
        if (a.rfind("--", 0) == 0) {
            size_t eq = a.find('=');
            name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { value = a.substr(eq + 1); have = true; }
        } else if (a.rfind("-o", 0) == 0) {
            name = "output_directory";
            if (a.size() > 2) { value = a.substr(a[2] == '=' ? 3 : 2); have = true; }
        } else {
            die(1, "error: Found argument '" + a + "' which wasn't expected, or isn't valid in this context\n\n" + USAGE);
        }
        if (std::find_if(std::begin(known), std::end(known), [&](const char *k) { return name == k; }) == std::end(known))
            die(1, "error: Found argument '" + a + "' which wasn't expected, or isn't valid in this context\n\n" + USAGE);
        if (!have) {
            if (i + 1 >= argc) die(1, "error: The argument '--" + name + " <" + name + ">' requires a value but none was supplied");
            value = argv[++i];
        }
        got[name] = value;
    }
    for (const char *req : {"alt1", "alt2", "barcodes1", "barcodes2", "num_cells_1", "output_directory", "ref1", "ref2", "vcf1", "vcf2"})
        if (!got.count(req)) die(1, std::string("error: The following required arguments were not provided:\n    --") + req + " <" + req + ">\n\n" + USAGE);
    Params p;
    p.vcf1 = got["vcf1"]; p.vcf2 = got["vcf2"]; p.alt1 = got["alt1"]; p.ref1 = got["ref1"]; p.alt2 = got["alt2"]; p.ref2 = got["ref2"];
    p.barcodes1 = got["barcodes1"]; p.barcodes2 = got["barcodes2"]; p.output_directory = got["output_directory"];
    p.num_cells_1 = parse_usize("num_cells_1", got["num_cells_1"]);
    if (got.count("num_cells_2")) p.num_cells_2 = parse_usize("num_cells_2", got["num_cells_2"]);
    if (got.count("dataset2_mask")) p.dataset2_mask = got["dataset2_mask"];
    if (got.count("seed")) p.seed = parse_usize("seed", got["seed"]);
    if (got.count("downsample_rate")) {
        char *end = nullptr;
        p.downsample_rate = strtod(got["downsample_rate"].c_str(), &end);
        if (got["downsample_rate"].empty() || *end) die(EXIT_PANIC, "cannot parse --downsample_rate");
    }
    return p;
}

// get_locus_mapping (main.rs:197-231): 1-based record numbers; returns dataset-2 locus -> output locus, and the output locus count
std::pair<std::unordered_map<uint64_t, uint64_t>, uint64_t> get_locus_mapping(const Params &p)
{
    std::map<std::pair<std::string, uint64_t>, uint64_t> chr_pos_to_locus;
    uint64_t record = 1;
    std::string line;
    {
        Lines in(p.vcf1);
        while (in.next(line)) {
            if (!line.empty() && line[0] == '#') continue;
            const size_t t1 = line.find('\t'), t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1);
            if (t1 == std::string::npos) die(EXIT_PANIC, "index out of bounds: vcf1 record without a POS column: " + line);
            const std::string pos = line.substr(t1 + 1, t2 == std::string::npos ? std::string::npos : t2 - t1 - 1);
            chr_pos_to_locus[{line.substr(0, t1), parse_usize("vcf1 POS", pos)}] = record;  // (a repeated position keeps the LAST record)
            record++;
        }
    }
    std::unordered_map<uint64_t, uint64_t> map2;
    uint64_t record2 = 1;
    Lines in(p.vcf2);
    while (in.next(line)) {
        if (!line.empty() && line[0] == '#') continue;
        const size_t t1 = line.find('\t'), t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1);
        if (t1 == std::string::npos) die(EXIT_PANIC, "index out of bounds: vcf2 record without a POS column: " + line);
        const std::string pos = line.substr(t1 + 1, t2 == std::string::npos ? std::string::npos : t2 - t1 - 1);
        auto it = chr_pos_to_locus.find({line.substr(0, t1), parse_usize("vcf2 POS", pos)});
        if (it != chr_pos_to_locus.end()) map2[record2] = it->second;
        else map2[record2] = record++;
        record2++;
    }
    return {map2, record - 1};
}

// consume_mtx_header (main.rs:283-299): three lines per file, dims from the REF file's third line
std::pair<uint64_t, uint64_t> consume_mtx_header(Lines &alt, Lines &ref)
{
    std::string la, lr;
    uint64_t loci = 0, cells = 0;
    for (int x = 0; x < 3; x++) {
        alt.next(la);
        ref.next(lr);
        if (x == 2) {
            auto t = tokens(lr);
            if (t.size() < 2) die(EXIT_PANIC, "index out of bounds: matrix market size line '" + lr + "'");
            loci = parse_usize("size line", t[0]);
            cells = parse_usize("size line", t[1]);
        }
    }
    return {loci, cells};
}

// select_cells (main.rs:246-255): a uniform sample without replacement of 1-based cell ids, in selection order (the reference's
// choose_multiple returns reservoir order, also not sorted) — partial Fisher-Yates on the tool's own stream
std::vector<uint64_t> select_cells(const Params &p, uint64_t want, uint64_t total)
{
    if (want > total) die(EXIT_PANIC, "cant ask for more cells than exist in dataset");
    Rng rng(p.seed);
    std::vector<uint64_t> ids(total);
    for (uint64_t i = 0; i < total; i++) ids[i] = i + 1;
    for (uint64_t i = 0; i < want; i++) std::swap(ids[i], ids[i + rng.below(total - i)]);
    ids.resize(want);
    return ids;
}

std::vector<std::string> read_lines(const std::string &path)
{
    std::vector<std::string> out;
    Lines in(path);
    std::string line;
    while (in.next(line)) out.push_back(line);
    return out;
}

using Entry = std::tuple<uint64_t, uint64_t, uint64_t, uint64_t>;  // locus, cell, ref, alt — the reference's sort key order

// one dataset's entries of the selected cells (main.rs:74-110)
void collect(const Params &p, const std::string &alt_path, const std::string &ref_path, const std::unordered_map<uint64_t, uint64_t> &cell_ids,
             const std::unordered_map<uint64_t, uint64_t> *locus_map, Rng &rng, std::vector<Entry> &lines)
{
    Lines alt(alt_path), ref(ref_path);
    consume_mtx_header(alt, ref);
    std::string la, lr;
    while (alt.next(la) && ref.next(lr)) {  // izip!: stops at the shorter file
        const auto ta = tokens(la), tr = tokens(lr);
        if (ta.size() < 3 || tr.size() < 3) die(EXIT_PANIC, "index out of bounds: mtx line '" + la + "' / '" + lr + "'");
        const uint64_t locus = parse_usize("mtx locus", ta[0]), cell = parse_usize("mtx cell", ta[1]);
        uint64_t alt_count = parse_usize("mtx count", ta[2]), ref_count = parse_usize("mtx count", tr[2]);
        auto it = cell_ids.find(cell);
        if (it == cell_ids.end()) continue;
        uint64_t out_locus = locus;
        if (locus_map) {
            auto lt = locus_map->find(locus);
            if (lt == locus_map->end()) die(EXIT_PANIC, "called `Option::unwrap()` on a `None` value: dataset 2 locus " + std::to_string(locus) + " has no vcf2 record");
            out_locus = lt->second;
        }
        const uint64_t r0 = ref_count, a0 = alt_count;  // every read is dropped independently; ref reads first (main.rs:82-87)
        for (uint64_t k = 0; k < r0; k++)
            if (rng.uniform() < p.downsample_rate) ref_count--;
        for (uint64_t k = 0; k < a0; k++)
            if (rng.uniform() < p.downsample_rate) alt_count--;
        lines.emplace_back(out_locus, it->second, ref_count, alt_count);
    }
}

FILE *create(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) die(EXIT_PANIC, "Unable to create file " + path);
    return f;
}

}  // namespace

int main(int argc, char **argv)
{
    const Params p = load_params(argc, argv);
    (void)mkdir(p.output_directory.c_str(), 0777);  // create_output_dir: non-recursive, failure ignored
    auto [locus2to1, total_loci_out] = get_locus_mapping(p);

    uint64_t total_cells_1 = 0, total_cells_2 = 0;
    { Lines a(p.alt1), r(p.ref1); total_cells_1 = consume_mtx_header(a, r).second; }
    const std::vector<uint64_t> cells1 = select_cells(p, p.num_cells_1, total_cells_1);
    { Lines a(p.alt2), r(p.ref2); total_cells_2 = consume_mtx_header(a, r).second; }
    const std::vector<std::string> barcodes1 = read_lines(p.barcodes1), barcodes2 = read_lines(p.barcodes2);
    std::vector<uint64_t> cells2;
    if (p.dataset2_mask) {  // select_cells_by_barcode (main.rs:257-281): dataset-2 cells whose barcode is listed, in file order
        const std::vector<std::string> mask = read_lines(*p.dataset2_mask);
        const std::unordered_set<std::string> wanted(mask.begin(), mask.end());
        for (size_t i = 0; i < barcodes2.size(); i++)
            if (wanted.count(barcodes2[i])) cells2.push_back(i + 1);
    } else {
        if (!p.num_cells_2) die(EXIT_PANIC, "missing argument num_cells_2 or dataset2_mask");
        cells2 = select_cells(p, *p.num_cells_2, total_cells_2);
    }
    const uint64_t n1 = p.num_cells_1, n2 = cells2.size();

    // get_cell_ids_and_output_barcodes (main.rs:141-188)
    std::unordered_map<uint64_t, uint64_t> ids1, ids2;
    {
        FILE *bc = create(p.output_directory + "/barcodes.tsv"), *gt = create(p.output_directory + "/gt.tsv");
        uint64_t out_id = 1;
        for (uint64_t c : cells1) {
            if (c - 1 >= barcodes1.size()) die(EXIT_PANIC, "index out of bounds: barcodes1 has " + std::to_string(barcodes1.size()) + " lines, cell " + std::to_string(c));
            ids1[c] = out_id++;
            fprintf(bc, "%s\n", barcodes1[c - 1].c_str());
            fprintf(gt, "%s\tmajority\n", barcodes1[c - 1].c_str());
        }
        for (uint64_t c : cells2) {
            if (c - 1 >= barcodes2.size()) die(EXIT_PANIC, "index out of bounds: barcodes2 has " + std::to_string(barcodes2.size()) + " lines, cell " + std::to_string(c));
            ids2[c] = out_id++;
            std::string b = barcodes2[c - 1];
            if (!b.empty()) b.pop_back();  // "...-1" -> "...-2": the two datasets may share barcodes
            b += '2';
            fprintf(bc, "%s\n", b.c_str());
            fprintf(gt, "%s\tminority\n", b.c_str());
        }
        fclose(bc);
        fclose(gt);
    }

    // output_new_mtxs (main.rs:52-116)
    std::vector<Entry> lines;
    Rng rng(p.seed);  // (a fresh stream, like the reference's second StdRng from the same seed)
    collect(p, p.alt1, p.ref1, ids1, nullptr, rng, lines);
    collect(p, p.alt2, p.ref2, ids2, &locus2to1, rng, lines);
    std::sort(lines.begin(), lines.end());
    FILE *fa = create(p.output_directory + "/alt.mtx"), *fr = create(p.output_directory + "/ref.mtx");
    for (FILE *f : {fa, fr})
        fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%% written by sprs\n%llu\t%llu\t%d\n", (unsigned long long)total_loci_out,
                (unsigned long long)(n1 + n2), 0);
    for (const Entry &e : lines) {
        fprintf(fa, "%llu\t%llu\t%llu\n", (unsigned long long)std::get<0>(e), (unsigned long long)std::get<1>(e), (unsigned long long)std::get<3>(e));
        fprintf(fr, "%llu\t%llu\t%llu\n", (unsigned long long)std::get<0>(e), (unsigned long long)std::get<1>(e), (unsigned long long)std::get<2>(e));
    }
    if (fclose(fa) != 0 || fclose(fr) != 0) die(EXIT_PANIC, "could not write the output matrices");
    printf("%llu,%llu\n", (unsigned long long)n1, (unsigned long long)n2);
    return 0;
}
