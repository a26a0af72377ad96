// Native input pipeline: threaded PNG decode straight into (pinned) uint8 NCHW batch buffers.
//
// Reference: every script reads `root/{id}.png` with torchvision.io.read_image (libpng, C++) inside
// torch DataLoader worker PROCESSES, one image per call, then collates and pickles the batch back to
// the trainer (single.py:45-65, ddp.py:343-360).  Here the loader is a C++ thread pool inside the
// training process: a batch is decoded by all threads at once directly into its final buffer (no
// per-image tensors, no collate copy, no inter-process pickling), `depth` batches ahead of the
// consumer, with the GIL released while waiting.  The decoder is self-contained on zlib (the image
// ships no libpng headers): 8-bit non-interlaced gray / RGB / palette / gray+alpha / RGBA, alpha
// dropped and gray replicated exactly as PIL's convert("RGB") does.  Anything else (16-bit, Adam7)
// is reported per slot as "unsupported" and decoded by the Python fallback (ddl/data.py).
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <torch/extension.h>

namespace {

struct PngError : std::runtime_error {
  bool unsupported;
  PngError(const std::string& m, bool unsup) : std::runtime_error(m), unsupported(unsup) {}
};

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

struct PngHeader {
  uint32_t w = 0, h = 0;
  int depth = 0, color = 0, interlace = 0;
};

std::vector<uint8_t> read_file(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw PngError("cannot open " + path, false);
  std::vector<uint8_t> buf;
  if (std::fseek(f, 0, SEEK_END) == 0) {
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n > 0) {
      buf.resize((size_t)n);
      if (std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) buf.clear();
    }
  }
  std::fclose(f);
  if (buf.empty()) throw PngError("cannot read " + path, false);
  return buf;
}

PngHeader parse_header(const std::vector<uint8_t>& d, const std::string& path) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (d.size() < 33 || std::memcmp(d.data(), sig, 8) != 0) throw PngError(path + ": not a PNG file", false);
  if (be32(d.data() + 8) != 13 || std::memcmp(d.data() + 12, "IHDR", 4) != 0) throw PngError(path + ": IHDR missing", false);
  PngHeader h;
  h.w = be32(d.data() + 16);
  h.h = be32(d.data() + 20);
  h.depth = d[24];
  h.color = d[25];
  h.interlace = d[28];
  if (d[26] != 0 || d[27] != 0) throw PngError(path + ": unknown compression / filter method", false);
  return h;
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Decode one file into out[3][H][W] (uint8, planes `plane` elements apart).
void decode_png(const std::string& path, uint8_t* out, int H, int W) {
  const std::vector<uint8_t> d = read_file(path);
  const PngHeader hd = parse_header(d, path);
  if ((int)hd.h != H || (int)hd.w != W)
    throw std::runtime_error(path + ": image is " + std::to_string(hd.w) + "x" + std::to_string(hd.h) + ", loader expects " +
                             std::to_string(W) + "x" + std::to_string(H));      // a configuration error, not a decode failure
  int ch;
  switch (hd.color) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: throw PngError(path + ": bad colour type", false);
  }
  if (hd.depth != 8 || hd.interlace != 0)
    throw PngError(path + ": only 8-bit non-interlaced PNGs are decoded natively", true);

  // walk the chunks: palette + concatenated IDAT stream
  std::vector<uint8_t> pal;
  std::vector<uint8_t> z;
  size_t pos = 8;
  bool end = false;
  while (!end) {
    if (pos + 12 > d.size()) throw PngError(path + ": truncated chunk", false);
    const uint32_t len = be32(d.data() + pos);
    if ((size_t)len > d.size() - pos - 12) throw PngError(path + ": truncated chunk", false);
    const uint8_t* type = d.data() + pos + 4;
    const uint8_t* body = type + 4;
    // CRC of the CRITICAL chunks only (IHDR / PLTE / IDAT / IEND: upper-case first letter) - libpng / PIL, which the
    // reference's read_image sits on, tolerate a damaged ancillary chunk (tEXt, pHYs, ...) and so does this decoder
    const bool critical = (type[0] & 0x20) == 0;
    if (critical && (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(body + len))
      throw PngError(path + ": CRC mismatch in a critical chunk", false);
    if (std::memcmp(type, "IDAT", 4) == 0) z.insert(z.end(), body, body + len);
    else if (std::memcmp(type, "PLTE", 4) == 0) pal.assign(body, body + len);
    else if (std::memcmp(type, "IEND", 4) == 0) end = true;
    pos += 12 + (size_t)len;
  }
  if (hd.color == 3 && pal.size() < 3) throw PngError(path + ": palette image without PLTE", false);

  const size_t row = (size_t)W * ch, stride = row + 1;
  std::vector<uint8_t> raw(stride * (size_t)H);
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit(&zs) != Z_OK) throw PngError(path + ": inflateInit failed", false);
  zs.next_in = z.data();
  zs.avail_in = (uInt)z.size();
  zs.next_out = raw.data();
  zs.avail_out = (uInt)raw.size();
  const int rc = inflate(&zs, Z_FINISH);
  const size_t got = raw.size() - zs.avail_out;
  inflateEnd(&zs);
  if (rc != Z_STREAM_END || got != raw.size())
    throw PngError(path + ": corrupt image data (inflate)", false);

  // undo the row filters in place (bpp = bytes per pixel)
  const int bpp = ch;
  const uint8_t* prev = nullptr;
  for (int y = 0; y < H; ++y) {
    uint8_t* line = raw.data() + stride * (size_t)y;
    const int ft = line[0];
    uint8_t* cur = line + 1;
    switch (ft) {
      case 0: break;
      case 1:
        for (size_t i = bpp; i < row; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        break;
      case 2:
        if (prev) for (size_t i = 0; i < row; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
        break;
      case 3:
        for (size_t i = 0; i < row; ++i) {
          const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
          cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < row; ++i) {
          const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
          const int c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
          cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
        }
        break;
      default: throw PngError(path + ": bad row filter", false);
    }
    prev = cur;
  }

  // interleaved pixels -> three planes
  const size_t plane = (size_t)H * W;
  uint8_t* r = out;
  uint8_t* g = out + plane;
  uint8_t* b = out + 2 * plane;
  const size_t npal = pal.size() / 3;
  for (int y = 0; y < H; ++y) {
    const uint8_t* px = raw.data() + stride * (size_t)y + 1;
    const size_t o = (size_t)y * W;
    switch (hd.color) {
      case 2:
      case 6:
        for (int x = 0; x < W; ++x) { r[o + x] = px[ch * x]; g[o + x] = px[ch * x + 1]; b[o + x] = px[ch * x + 2]; }
        break;
      case 0:
      case 4:
        for (int x = 0; x < W; ++x) { const uint8_t v = px[ch * x]; r[o + x] = v; g[o + x] = v; b[o + x] = v; }
        break;
      default:   // palette
        for (int x = 0; x < W; ++x) {
          const size_t k = px[x];
          if (k >= npal) throw PngError(path + ": palette index out of range", false);
          r[o + x] = pal[3 * k]; g[o + x] = pal[3 * k + 1]; b[o + x] = pal[3 * k + 2];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Thread pool + batch pipeline
// ---------------------------------------------------------------------------------------------
struct Batch {
  at::Tensor images;                 // uint8 [m,3,H,W]
  at::Tensor index;                  // int64 [m]: dataset indices of the rows
  std::atomic<int> remaining{0};
  std::mutex mu;                     // guards error / unsupported
  std::string error;
  std::vector<int64_t> unsupported;  // slots left for the Python fallback decoder
};

struct Task {
  std::shared_ptr<Batch> batch;
  int slot;
  int64_t file;
};

class PngLoader {
 public:
  PngLoader(std::vector<std::string> paths, int64_t H, int64_t W, int64_t threads, int64_t depth, bool pin)
      : paths_(std::move(paths)), H_((int)H), W_((int)W), depth_((int)std::max<int64_t>(depth, 1)), pin_(pin) {
    TORCH_CHECK(H > 0 && W > 0, "PngLoader: bad image size");
    const int n = (int)std::max<int64_t>(threads, 1);
    for (int i = 0; i < n; ++i) workers_.emplace_back([this] { work(); });
  }

  ~PngLoader() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      tasks_.clear();
    }
    cv_task_.notify_all();
    for (auto& t : workers_) t.join();
  }

  int64_t size() const { return (int64_t)paths_.size(); }

  // Begin an epoch: `order` = dataset indices in visiting order; batches of `batch` (the tail kept
  // unless drop_last).  Batches still pending from an abandoned epoch are dropped.
  void start_epoch(std::vector<int64_t> order, int64_t batch, bool drop_last) {
    TORCH_CHECK(batch > 0, "PngLoader: batch size must be positive");
    for (int64_t i : order) TORCH_CHECK(i >= 0 && i < size(), "PngLoader: index ", i, " out of range");
    {
      std::lock_guard<std::mutex> lk(mu_);
      tasks_.clear();
    }
    drain_pending();
    order_ = std::move(order);
    batch_ = batch;
    const int64_t n = (int64_t)order_.size();
    nbatches_ = drop_last ? n / batch : (n + batch - 1) / batch;
    issued_ = 0;
    for (int i = 0; i < depth_; ++i) issue();
  }

  int64_t batches_left() const { return nbatches_ - issued_ + (int64_t)pending_.size(); }

  // Next decoded batch: (images, dataset indices, slots the native decoder does not support).  Call with the
  // GIL released (the binding does); throws on unreadable / corrupt files.
  std::tuple<at::Tensor, at::Tensor, std::vector<int64_t>> next() {
    TORCH_CHECK(!pending_.empty(), "PngLoader: no batch pending (epoch exhausted)");
    std::shared_ptr<Batch> b = pending_.front();
    pending_.pop_front();
    issue();                                          // keep `depth` batches in flight
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [&] { return b->remaining.load() == 0; });
    }
    if (!b->error.empty()) throw std::runtime_error(b->error);
    return {b->images, b->index, b->unsupported};
  }

 private:
  void issue() {
    if (issued_ >= nbatches_) return;
    const int64_t a = issued_ * batch_;
    const int64_t m = std::min<int64_t>(batch_, (int64_t)order_.size() - a);
    auto b = std::make_shared<Batch>();
    b->images = at::empty({m, 3, H_, W_}, at::TensorOptions().dtype(at::kByte).pinned_memory(pin_));
    b->index = at::empty({m}, at::TensorOptions().dtype(at::kLong));
    std::memcpy(b->index.data_ptr<int64_t>(), order_.data() + a, sizeof(int64_t) * (size_t)m);
    b->remaining.store((int)m);
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int s = 0; s < (int)m; ++s) tasks_.push_back(Task{b, s, order_[a + s]});
    }
    cv_task_.notify_all();
    pending_.push_back(std::move(b));
    ++issued_;
  }

  void drain_pending() {      // the queue is empty: wait for the slots a worker had already picked up
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [&] { return running_ == 0; });
    }
    pending_.clear();
  }

  void work() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_task_.wait(lk, [&] { return stop_ || !tasks_.empty(); });
        if (stop_) return;
        t = std::move(tasks_.front());
        tasks_.pop_front();
        ++running_;
      }
      Batch& b = *t.batch;
      uint8_t* dst = b.images.data_ptr<uint8_t>() + (size_t)t.slot * 3 * (size_t)H_ * (size_t)W_;
      try {
        decode_png(paths_[(size_t)t.file], dst, H_, W_);
      } catch (const PngError& e) {
        // EVERY decode failure of a slot goes to the per-slot Python fallback (PIL, ddl/data.py) - a file the
        // reference's torchvision.io.read_image (libpng) accepts must never abort a batch here just because this
        // decoder is stricter; the fallback raises if PIL cannot read the file either.  I/O errors and a wrong image
        // size (std::runtime_error below) stay fatal.
        std::lock_guard<std::mutex> lk(b.mu);
        b.unsupported.push_back(t.slot);
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(b.mu);
        if (b.error.empty()) b.error = e.what();
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        --running_;
        b.remaining.fetch_sub(1);
      }
      cv_done_.notify_all();
    }
  }

  std::vector<std::string> paths_;
  int H_, W_, depth_;
  bool pin_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_task_, cv_done_;
  std::deque<Task> tasks_;
  int running_ = 0;
  bool stop_ = false;
  // consumer-side state (one consumer thread)
  std::vector<int64_t> order_;
  int64_t batch_ = 1, nbatches_ = 0, issued_ = 0;
  std::deque<std::shared_ptr<Batch>> pending_;
};

// one-shot helpers ----------------------------------------------------------------------------
at::Tensor decode_png_file(const std::string& path) {
  const std::vector<uint8_t> d = read_file(path);
  const PngHeader h = parse_header(d, path);
  auto out = at::empty({3, (int64_t)h.h, (int64_t)h.w}, at::TensorOptions().dtype(at::kByte));
  decode_png(path, out.data_ptr<uint8_t>(), (int)h.h, (int)h.w);
  return out;
}

std::tuple<int64_t, int64_t, int64_t, int64_t, int64_t> png_info(const std::string& path) {
  const std::vector<uint8_t> d = read_file(path);
  const PngHeader h = parse_header(d, path);
  return {(int64_t)h.w, (int64_t)h.h, (int64_t)h.depth, (int64_t)h.color, (int64_t)h.interlace};
}

}  // namespace

void register_io(pybind11::module_& m) {
  namespace py = pybind11;
  m.def("png_info", &png_info, "(width, height, bit depth, colour type, interlace) of a PNG file");
  m.def("decode_png", &decode_png_file, py::call_guard<py::gil_scoped_release>(),
        "decode an 8-bit non-interlaced PNG to a uint8 [3,H,W] tensor");
  py::class_<PngLoader>(m, "PngLoader")
      .def(py::init<std::vector<std::string>, int64_t, int64_t, int64_t, int64_t, bool>(), py::arg("paths"),
           py::arg("height"), py::arg("width"), py::arg("threads") = 4, py::arg("depth") = 2, py::arg("pin") = false)
      .def("start_epoch", &PngLoader::start_epoch, py::arg("order"), py::arg("batch"), py::arg("drop_last"),
           py::call_guard<py::gil_scoped_release>())
      .def("next", &PngLoader::next, py::call_guard<py::gil_scoped_release>())
      .def("batches_left", &PngLoader::batches_left)
      .def("__len__", &PngLoader::size);
}
