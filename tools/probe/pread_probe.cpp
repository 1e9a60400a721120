// Host-side probe: how fast do N threads pread() 256 MB pieces of one or two page-cache-resident files (tmpfs)?
// usage: pread_probe <threads per file> <file> [<file2>]     (the upload path of csrc/kernels_parse.hip does exactly this)
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static void rd(int fd, size_t off, size_t len, char *dst)
{
    while (len) {
        const ssize_t g = pread(fd, dst, len, (off_t)off);
        if (g <= 0) break;
        dst += g; off += (size_t)g; len -= (size_t)g;
    }
}
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int nthr = atoi(argv[1]), nfiles = argc - 2;
    const size_t piece = 256ull << 20;
    std::vector<int> fds;
    std::vector<size_t> sizes;
    for (int i = 0; i < nfiles; i++) {
        const int fd = open(argv[2 + i], O_RDONLY);
        if (fd < 0) return 3;
        fds.push_back(fd);
        sizes.push_back((size_t)lseek(fd, 0, SEEK_END));
    }
    std::vector<char *> bufs;
    for (int f = 0; f < nfiles; f++) { char *b = (char *)malloc(piece); memset(b, 1, piece); bufs.push_back(b); }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> files;
    size_t total = 0;
    for (int f = 0; f < nfiles; f++) {
        total += sizes[f];
        files.emplace_back([&, f] {
            for (size_t o = 0; o < sizes[f]; o += piece) {
                const size_t len = std::min(piece, sizes[f] - o), sl = (len + nthr - 1) / nthr;
                std::vector<std::thread> th;
                for (int t = 0; t < nthr; t++)
                    th.emplace_back([&, t] {
                        const size_t b0 = std::min(len, (size_t)t * sl), b1 = std::min(len, b0 + sl);
                        if (b1 > b0) rd(fds[f], o + b0, b1 - b0, bufs[f] + b0);
                    });
                for (auto &x : th) x.join();
            }
        });
    }
    for (auto &x : files) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d threads x %d file(s): %.1f GB/s\n", nthr, nfiles, total / s / 1e9);
    return 0;
}
