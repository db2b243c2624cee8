// Sequence driver in the shape of Examples/Monocular/mono_tum.cc of the
// reference: reads <sequence>/rgb.txt (:74-114 there: three header lines, then
// "timestamp filename" per line) and the ORBextractor.* / Camera.RGB entries of
// a settings file such as Examples/Monocular/TUM1.yaml, streams the frames in
// batches through liborbhip (orb_extract_batch) and writes every frame's
// keypoints and descriptors to one binary file.
//
//   orb_tum path_to_settings path_to_sequence out.orbk [batch] [--list-only]
//
// Images: binary PGM (P5, maxval 255) or PPM (P6, maxval 255).  Colour images
// are converted as Tracking::GrabImageMonocular does (src/Tracking.cc:189-197):
// cv::imread hands out B,G,R order, and Camera.RGB chooses RGB2GRAY or BGR2GRAY,
// 8-bit fixed point (R*4899 + G*9617 + B*1868 + 8192) >> 14.
//
// Output: "ORBK" u32 version(1) u32 nframes, then per frame
//   f64 timestamp, u32 width, u32 height, u32 n, n x orb_keypoint (28 B), n x 32 B.
//
// While the device extracts one batch a host thread reads the next one.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/orb_hip.h"

using namespace std;

// Examples/Monocular/mono_tum.cc:74-114
static void LoadImages(const string& strFile, vector<string>& vstrImageFilenames, vector<double>& vTimestamps) {
    ifstream f(strFile.c_str());
    string s0;
    getline(f, s0);
    getline(f, s0);
    getline(f, s0);
    while (!f.eof()) {
        string s;
        getline(f, s);
        if (!s.empty()) {
            stringstream ss;
            ss << s;
            double t;
            string sRGB;
            ss >> t;
            vTimestamps.push_back(t);
            ss >> sRGB;
            vstrImageFilenames.push_back(sRGB);
        }
    }
}

// "Key: value" lines of an OpenCV FileStorage YAML file; returns false when the key is absent.
static bool ReadSetting(const string& path, const string& key, double& value) {
    ifstream f(path.c_str());
    string line;
    while (getline(f, line)) {
        const size_t c = line.find(':');
        if (c == string::npos || line[0] == '#' || line[0] == '%') continue;
        string k = line.substr(0, c);
        k.erase(0, k.find_first_not_of(" \t"));
        k.erase(k.find_last_not_of(" \t") + 1);
        if (k != key) continue;
        value = atof(line.substr(c + 1).c_str());
        return true;
    }
    return false;
}

struct Gray {
    int w = 0, h = 0;
    vector<uint8_t> px;
};

static bool NextToken(FILE* fp, string& tok) {  // PNM header token, '#' comments skipped
    tok.clear();
    int c;
    while ((c = fgetc(fp)) != EOF) {
        if (c == '#') {
            while ((c = fgetc(fp)) != EOF && c != '\n') {
            }
            continue;
        }
        if (!isspace(c)) break;
    }
    if (c == EOF) return false;
    do {
        tok.push_back((char)c);
        c = fgetc(fp);
    } while (c != EOF && !isspace(c));
    return true;
}

static bool LoadGray(const string& path, bool rgbFlag, Gray& out) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    string magic, sw, sh, smax;
    bool ok = NextToken(fp, magic) && NextToken(fp, sw) && NextToken(fp, sh) && NextToken(fp, smax);
    const int w = ok ? atoi(sw.c_str()) : 0, h = ok ? atoi(sh.c_str()) : 0;
    ok = ok && w > 0 && h > 0 && atoi(smax.c_str()) == 255 && (magic == "P5" || magic == "P6");
    if (ok) {
        out.w = w;
        out.h = h;
        out.px.resize((size_t)w * h);
        if (magic == "P5") {
            ok = fread(out.px.data(), 1, out.px.size(), fp) == out.px.size();
        } else {
            vector<uint8_t> rgb((size_t)w * h * 3);
            ok = fread(rgb.data(), 1, rgb.size(), fp) == rgb.size();
            for (size_t i = 0; ok && i < out.px.size(); i++) {
                // memory order after cv::imread is B,G,R; RGB2GRAY weighs channel 0 as red
                const int c0 = rgb[3 * i + 2], c1 = rgb[3 * i + 1], c2 = rgb[3 * i + 0];
                const int r = rgbFlag ? c0 : c2, g = c1, b = rgbFlag ? c2 : c0;
                out.px[i] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
            }
        }
    }
    fclose(fp);
    return ok;
}

int main(int argc, char const* argv[]) {
    if (argc < 4) {
        cerr << endl << "Usage: ./orb_tum path_to_settings path_to_sequence out.orbk [batch] [--list-only]" << endl;
        return 1;
    }
    bool listOnly = false;
    int batch = 32;
    for (int i = 4; i < argc; i++) {
        if (string(argv[i]) == "--list-only") listOnly = true;
        else batch = max(1, atoi(argv[i]));
    }
    vector<string> vstrImageFilenames;
    vector<double> vTimestamps;
    const string strFile = string(argv[2]) + "/rgb.txt";
    LoadImages(strFile, vstrImageFilenames, vTimestamps);
    const int nImages = (int)vstrImageFilenames.size();

    // src/Tracking.cc:105-109 reads these five keys, :97 Camera.RGB
    double nFeatures = 0, fScaleFactor = 0, nLevels = 0, fIniThFAST = 0, fMinThFAST = 0, nRGB = 0;
    const string settings = argv[1];
    if (!ReadSetting(settings, "ORBextractor.nFeatures", nFeatures) || !ReadSetting(settings, "ORBextractor.scaleFactor", fScaleFactor) ||
        !ReadSetting(settings, "ORBextractor.nLevels", nLevels) || !ReadSetting(settings, "ORBextractor.iniThFAST", fIniThFAST) ||
        !ReadSetting(settings, "ORBextractor.minThFAST", fMinThFAST)) {
        cerr << "Failed to read the ORBextractor.* settings from " << settings << endl;
        return 1;
    }
    ReadSetting(settings, "Camera.RGB", nRGB);

    cout << endl << "ORB Extractor Parameters: " << endl;
    cout << "- Number of Features: " << (int)nFeatures << endl;
    cout << "- Scale Levels: " << (int)nLevels << endl;
    cout << "- Scale Factor: " << fScaleFactor << endl;
    cout << "- Initial Fast Threshold: " << (int)fIniThFAST << endl;
    cout << "- Minimum Fast Threshold: " << (int)fMinThFAST << endl;
    cout << endl << "-------" << endl;
    cout << "Start processing sequence ..." << endl;
    cout << "Images in the sequence: " << nImages << endl << endl;
    if (listOnly) return 0;
    if (nImages == 0) {
        cerr << "No images listed in " << strFile << endl;
        return 1;
    }

    Gray first;
    if (!LoadGray(string(argv[2]) + "/" + vstrImageFilenames[0], nRGB != 0, first)) {
        cerr << endl << "Failed to load image at: " << argv[2] << "/" << vstrImageFilenames[0] << endl;
        return 1;
    }
    unsigned quirks = ORB_QUIRKS_FORK;
    if (const char* e = getenv("ORBHIP_QUIRKS")) quirks = (unsigned)strtoul(e, nullptr, 0);
    orb_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.nfeatures = (int)nFeatures;
    prm.scale_factor = (float)fScaleFactor;
    prm.nlevels = (int)nLevels;
    prm.ini_th_fast = (int)fIniThFAST;
    prm.min_th_fast = (int)fMinThFAST;
    prm.max_width = first.w;
    prm.max_height = first.h;
    prm.max_batch = batch;
    prm.device = getenv("ORBHIP_DEVICE") ? atoi(getenv("ORBHIP_DEVICE")) : 0;
    prm.quirks = quirks;
    orb_ctx* ctx = nullptr;
    if (orb_create(&prm, &ctx) != ORB_OK) {
        cerr << "orb_create: " << orb_last_error() << endl;
        return 1;
    }
    const int cap = orb_max_keypoints_per_frame(ctx);

    FILE* out = fopen(argv[3], "wb");
    if (!out) {
        cerr << "Cannot open " << argv[3] << " for writing" << endl;
        return 1;
    }
    const uint32_t header[3] = {0x4B42524Fu /* "ORBK" */, 1u, (uint32_t)nImages};
    fwrite(header, 4, 3, out);

    // loader: fills one of two frame sets while the other is on the device
    struct Set {
        vector<Gray> frames;
        int first = 0, count = 0;
        bool ok = true;
        string failed;
    } sets[2];
    auto load = [&](Set& s, int firstIdx) {
        s.first = firstIdx;
        s.count = max(0, min(batch, nImages - firstIdx));
        s.frames.resize(s.count);
        s.ok = true;
        for (int i = 0; i < s.count; i++) {
            const string p = string(argv[2]) + "/" + vstrImageFilenames[firstIdx + i];
            if (!LoadGray(p, nRGB != 0, s.frames[i]) || s.frames[i].w != first.w || s.frames[i].h != first.h) {
                s.ok = false;
                s.failed = p;
                return;
            }
        }
    };
    vector<orb_keypoint> kps((size_t)batch * cap);
    vector<uint8_t> desc((size_t)batch * cap * 32);
    vector<int> counts(batch);
    vector<float> vTimesTrack;
    load(sets[0], 0);
    int cur = 0, rc = 0;
    size_t totalKeys = 0;
    for (int base = 0; base < nImages && rc == 0; base += batch, cur ^= 1) {
        Set& s = sets[cur];
        if (!s.ok) {
            cerr << endl << "Failed to load image at: " << s.failed << endl;
            rc = 1;
            break;
        }
        thread next([&, base]() { load(sets[cur ^ 1], base + batch); });
        vector<const uint8_t*> ptrs(s.count);
        for (int i = 0; i < s.count; i++) ptrs[i] = s.frames[i].px.data();
        const auto t1 = chrono::steady_clock::now();
        if (orb_extract_batch(ctx, ptrs.data(), s.count, first.w, first.h, first.w, kps.data(), desc.data(), cap, counts.data()) != ORB_OK) {
            cerr << "orb_extract_batch: " << orb_last_error() << endl;
            rc = 1;
        }
        const auto t2 = chrono::steady_clock::now();
        const double ttrack = chrono::duration_cast<chrono::duration<double> >(t2 - t1).count();
        for (int i = 0; i < s.count && rc == 0; i++) {
            vTimesTrack.push_back((float)(ttrack / s.count));
            const double ts = vTimestamps[s.first + i];
            const uint32_t meta[3] = {(uint32_t)first.w, (uint32_t)first.h, (uint32_t)counts[i]};
            fwrite(&ts, 8, 1, out);
            fwrite(meta, 4, 3, out);
            fwrite(&kps[(size_t)i * cap], sizeof(orb_keypoint), counts[i], out);
            fwrite(&desc[(size_t)i * cap * 32], 32, counts[i], out);
            totalKeys += counts[i];
        }
        next.join();
    }
    fclose(out);
    orb_destroy(ctx);
    if (rc) return rc;

    // statistics in the form of upstream mono_tum.cc
    sort(vTimesTrack.begin(), vTimesTrack.end());
    float totaltime = 0;
    for (size_t ni = 0; ni < vTimesTrack.size(); ni++) totaltime += vTimesTrack[ni];
    cout << "-------" << endl << endl;
    cout << "frames: " << vTimesTrack.size() << ", keypoints: " << totalKeys << endl;
    cout << "median extraction time per frame: " << vTimesTrack[vTimesTrack.size() / 2] << endl;
    cout << "mean extraction time per frame: " << totaltime / vTimesTrack.size() << endl;
    return 0;
}
