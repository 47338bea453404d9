// cv_compat.hpp -- the handful of OpenCV value types the reference's Camera/Matcher/VISystem surface
// uses, so the adapter classes build where OpenCV is absent (this image, the GPU box).  Field order
// and sizes match OpenCV's (cv::KeyPoint 28 B == vis_keypoint, cv::DMatch 16 B == vis_dmatch); on a
// machine with OpenCV, define VISLAM_USE_OPENCV and the real headers are used instead.
#ifndef VISLAM_CV_COMPAT_HPP_
#define VISLAM_CV_COMPAT_HPP_
#ifdef VISLAM_USE_OPENCV
#include <opencv2/core.hpp>
#else
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
namespace cv {
typedef std::string String;
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f; typedef Point_<double> Point2d;
template <class T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
    template <class U> Point3_(const Point3_<U>& o) : x((T)o.x), y((T)o.y), z((T)o.z) {}
};
template <class T> inline Point3_<T> operator+(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> inline Point3_<T> operator-(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> inline Point3_<T> operator-(const Point3_<T>& a) { return Point3_<T>(-a.x, -a.y, -a.z); }
template <class T> inline Point3_<T> operator*(const Point3_<T>& a, double s) { return Point3_<T>((T)(a.x * s), (T)(a.y * s), (T)(a.z * s)); }
typedef Point3_<float> Point3f; typedef Point3_<double> Point3d;
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id;
                  KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {} };
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance;
                DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
                DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
                bool operator<(const DMatch& m) const { return distance < m.distance; } };
static_assert(sizeof(KeyPoint) == 28 && sizeof(DMatch) == 16, "layout must match the C ABI");
enum { CV_8U = 0, CV_16S = 3, CV_32F = 5, CV_8UC1 = 0, CV_16SC1 = 3, CV_32FC1 = 5 };               // OpenCV depth codes
// minimal single-channel matrix with shared storage (enough for grayImage / descriptors / gradients / point lists / K)
struct Mat {
    int rows = 0, cols = 0, depth = CV_8U; size_t step = 0; uint8_t* data = nullptr;
    std::shared_ptr<std::vector<uint8_t>> store;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    static size_t esz(int type) { return type == CV_16S ? 2 : (type == CV_32F ? 4 : 1); }
    size_t elemSize() const { return esz(depth); }
    void create(int r, int c, int type) { rows = r; cols = c; depth = type; step = (size_t)c * esz(type); store = std::make_shared<std::vector<uint8_t>>((size_t)r * step); data = store->data(); }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type) { Mat m(r, c, type); if (type == CV_32F) for (int i = 0; i < r && i < c; i++) m.at<float>(i, i) = 1.f; return m; }
    bool empty() const { return rows == 0 || cols == 0; }
    void release() { rows = cols = 0; step = 0; data = nullptr; store.reset(); }
    Mat clone() const { Mat m; if (!empty()) { m.create(rows, cols, depth); for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * esz(depth)); } return m; }
    void copyTo(Mat& m) const { m = clone(); }
    Mat rowRange(int a, int b) const { Mat m = *this; m.data = data + (size_t)a * step; m.rows = b - a; return m; }
    template <class T> T& at(int y, int x) { return *reinterpret_cast<T*>(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <class T> const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + (size_t)y * step + (size_t)x * sizeof(T)); }
};
// cv::Matx33f: 3x3 float value matrix, row-major `val`
struct Matx33f {
    float val[9];
    Matx33f() { for (int i = 0; i < 9; i++) val[i] = 0.f; }
    float& operator()(int r, int c) { return val[3 * r + c]; }
    const float& operator()(int r, int c) const { return val[3 * r + c]; }
    Matx33f t() const { Matx33f m; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m(r, c) = (*this)(c, r); return m; }
    static Matx33f eye() { Matx33f m; m(0, 0) = m(1, 1) = m(2, 2) = 1.f; return m; }
};
inline Matx33f operator*(const Matx33f& a, const Matx33f& b) {
    Matx33f m;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m(r, c) = a(r, 0) * b(0, c) + a(r, 1) * b(1, c) + a(r, 2) * b(2, c);
    return m;
}
template <class T> using Ptr = std::shared_ptr<T>;
// cv::CommandLineParser for the keys string of src/main_vi_slamGPU.cpp:26-39,50-54: "{name alias ... | default | help}" groups; arguments
// "-name=value" / "--name=value", a bare "-name" is a flag.  has(name): the argument was given (or the key has a non-empty default);
// get<T>(name): the given value, else the default, through operator>>.
class CommandLineParser {
public:
    CommandLineParser(int argc, const char* const argv[], const String& keys) {
        size_t p = 0;
        while ((p = keys.find('{', p)) != String::npos) {
            const size_t e = keys.find('}', p);
            if (e == String::npos) break;
            const String body = keys.substr(p + 1, e - p - 1);
            const size_t b1 = body.find('|'), b2 = b1 == String::npos ? String::npos : body.find('|', b1 + 1);
            Key k; k.def = b1 == String::npos ? String() : trim(body.substr(b1 + 1, b2 == String::npos ? String::npos : b2 - b1 - 1));
            String names = body.substr(0, b1), n;
            for (size_t i = 0; i <= names.size(); i++) {
                if (i == names.size() || names[i] == ' ' || names[i] == '\t') { if (!n.empty()) k.names.push_back(n); n.clear(); }
                else n += names[i];
            }
            keys_.push_back(k);
            p = e + 1;
        }
        for (int i = 1; i < argc; i++) {
            String a = argv[i];
            if (a.empty() || a[0] != '-') continue;
            a = a.substr(a.size() > 1 && a[1] == '-' ? 2 : 1);
            const size_t eq = a.find('=');
            const String name = a.substr(0, eq), val = eq == String::npos ? String("true") : a.substr(eq + 1);
            for (auto& k : keys_) for (auto& nm : k.names) if (nm == name) { k.given = true; k.val = val; }
        }
    }
    bool has(const String& name) const { const Key* k = find(name); return k && (k->given || !k->def.empty()); }
    template <class T> T get(const String& name) const {
        const Key* k = find(name);
        const String v = k ? (k->given ? k->val : k->def) : String();
        return convert<T>(v);
    }
    bool check() const { return true; }
private:
    struct Key { std::vector<String> names; String def, val; bool given = false; };
    std::vector<Key> keys_;
    const Key* find(const String& name) const { for (auto& k : keys_) for (auto& nm : k.names) if (nm == name) return &k; return nullptr; }
    static String trim(const String& s) { const size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t"); return a == String::npos ? String() : s.substr(a, b - a + 1); }
    template <class T> static T convert(const String& v);
};
template <> inline String CommandLineParser::convert<String>(const String& v) { return v; }
template <> inline int CommandLineParser::convert<int>(const String& v) { return v.empty() ? 0 : std::atoi(v.c_str()); }
template <> inline double CommandLineParser::convert<double>(const String& v) { return v.empty() ? 0.0 : std::atof(v.c_str()); }
}  // namespace cv
#endif
#endif
