// Units of cluster_kernel compiled at run time for the exact constraint-type set of a scene (round 6).
// A prebuilt unit carries the code of every type of its family (contacts: 8, hot: 16, wide: all 44), and the code a scene never runs still costs the types it does run
// registers and scheduling: the headline scene is 9 % slower on the wide unit than on the hot one, the wide 1024-thread unit spills 681 VGPRs where a unit for the
// sixteen hot types + seven widened joint types spills 43 (profiles/r06_s22_ab_family_headline.txt, r06_s23_ab_rigs_mask.txt). The reference's counterpart is its
// per-type registration: a batch runs the TypeProcessors of the types it holds and nothing else (BepuPhysics/DefaultTypes.cs:18-63).
// A specialised unit is bepu_cluster_variant.inc compiled with -DBEPU_VARIANT_TYPE_MASK=<bit per type id> — the switch cases of the other types are left out, nothing
// else differs: same bits by construction (tests/test_gpu_type_families.py asserts it) — by hipcc as a child process on a host thread of the library, into a shared
// object of its own in a cache directory (key: the sources' hash, the mask, the register budget, the plan kind); the object registers its kernels with the HIP runtime
// when it is dlopen'ed, so they are launched like the prebuilt ones. Until a unit is there — and wherever there is no compiler or no sources — the scene runs the
// nearest prebuilt family. The .so files of a cache can be shipped (bepuphysics2_amd/build.py prebuilds the units of the BASELINE.json scenes into csrc/units/).
#pragma once
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <fcntl.h>
#include <signal.h>
#include <condition_variable>

extern char** environ;

enum { kUnitUnavailable = 0, kUnitCompiling = 1, kUnitLoaded = 2, kUnitFailed = 3 };

struct SpecialUnit {
    std::atomic<int> state{kUnitCompiling};
    const void* kernel = nullptr;  // cluster_kernel<THREADS, false, WIDE, SHARED> of the object
    const void* traced = nullptr;
    std::string path, why;
    std::thread worker;
    std::mutex join_mutex;
    bool compiled_now = false;     // (false: found in the cache)
    double seconds = 0.0;
};

static std::string unit_source_dir() {
    static const std::string dir = [] {
        std::string d;
        if (const char* forced = getenv("BEPUHIP_UNIT_SOURCES")) d = forced;
        else {
            Dl_info info;
            if (dladdr((const void*)&unit_source_dir, &info) && info.dli_fname) { d = info.dli_fname; const size_t slash = d.rfind('/'); d = slash == std::string::npos ? "." : d.substr(0, slash); }
        }
        return (!d.empty() && access((d + "/bepu_cluster_variant.inc").c_str(), R_OK) == 0 && access((d + "/bepu_cluster_kernel.h").c_str(), R_OK) == 0) ? d : std::string();
    }();
    return dir;
}
static std::string unit_compiler() {
    static const std::string cc = [] {
        if (const char* forced = getenv("BEPUHIP_HIPCC")) return std::string(access(forced, X_OK) == 0 ? forced : "");
        for (const char* cand : {"/opt/rocm/bin/hipcc", "/usr/bin/hipcc", "/usr/local/bin/hipcc"}) if (access(cand, X_OK) == 0) return std::string(cand);
        return std::string();
    }();
    return cc;
}
static bool unit_make_dirs(const std::string& dir) {
    for (size_t at = 1; at <= dir.size(); ++at)
        if (at == dir.size() || dir[at] == '/') { const std::string part = dir.substr(0, at); if (mkdir(part.c_str(), 0755) != 0 && errno != EEXIST) return false; }
    return access(dir.c_str(), W_OK) == 0;
}
// Where the objects live: BEPUHIP_UNIT_CACHE, else units/ next to the sources (ships with the tree), else ~/.cache/bepuhip/units. `read_dirs`: where an object may be found.
static std::string unit_cache_dir(std::vector<std::string>* read_dirs = nullptr) {
    std::vector<std::string> dirs;
    if (const char* forced = getenv("BEPUHIP_UNIT_CACHE")) dirs.push_back(forced);
    if (!unit_source_dir().empty()) dirs.push_back(unit_source_dir() + "/units");
    if (const char* home = getenv("HOME")) dirs.push_back(std::string(home) + "/.cache/bepuhip/units");
    if (read_dirs) *read_dirs = dirs;
    for (auto& d : dirs) if (unit_make_dirs(d)) return d;
    return std::string();
}
// What a unit is compiled from, hashed: an object of another source state is never loaded.
static const char* const kUnitSources[] = {"bepu_cluster_variant.inc", "bepu_cluster_kernel.h", "bepu_kernels_common.h", "bepu_batch_kernels.h", "bepu_device_constraints.h", "bepu_device_math.h", "bepu_device_bounds.h"};
// The flags of bepuphysics2_amd/build.py (HIP_COMPILE_FLAGS; tests/test_unit_cache.py compares the two lists)
static const char* const kUnitFlags[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Xarch_device", "-fno-slp-vectorize", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-Wno-array-bounds"};
static uint64_t unit_sources_hash() {
    static const uint64_t hash = [] {
        uint64_t h = 1469598103934665603ull;
        for (const char* name : kUnitSources) {
            FILE* f = fopen((unit_source_dir() + "/" + name).c_str(), "rb");
            if (!f) return (uint64_t)0;
            char buffer[65536];
            for (size_t n; (n = fread(buffer, 1, sizeof(buffer), f)) > 0;) for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)buffer[i]; h *= 1099511628211ull; }
            fclose(f);
        }
        for (const char* flag : kUnitFlags) for (const char* p = flag; *p; ++p) { h ^= (unsigned char)*p; h *= 1099511628211ull; }
        return h ? h : (uint64_t)1;
    }();
    return hash;
}
struct UnitKey { unsigned long long mask; int budget; bool shared; };  // budget: the launch bound the unit is compiled for (1024 / 768 / 512 threads: 128 / 168 / 256 VGPRs)
constexpr unsigned long long kContactsTypeMask = 0xFFull, kHotTypeMask = 0xFFull | (1ull << kBallSocket) | (1ull << kAngularHinge) | (1ull << kSwingLimit) | (1ull << kTwistServo) |
                                                 (1ull << kTwistLimit) | (1ull << kAngularMotor) | (1ull << kSwivelHinge) | (1ull << kHinge);
static std::string unit_file_name(const UnitKey& key) {
    char name[160];
    snprintf(name, sizeof(name), "unit_%016llx_m%014llx_t%d%s.so", (unsigned long long)unit_sources_hash(), key.mask, key.budget, key.shared ? "s" : "");
    return name;
}
static std::vector<std::string> unit_compile_command(const UnitKey& key, const std::string& out) {
    std::vector<std::string> argv = {unit_compiler()};
    for (const char* flag : kUnitFlags) argv.push_back(flag);
    char define[96];
    argv.push_back("-shared");
    snprintf(define, sizeof(define), "-DBEPU_VARIANT_THREADS=%d", key.budget); argv.push_back(define);
    argv.push_back((key.mask & ~kHotTypeMask) ? "-DBEPU_VARIANT_WIDE=1" : "-DBEPU_VARIANT_WIDE=0");
    if (!(key.mask & ~kContactsTypeMask)) argv.push_back("-DBEPU_VARIANT_CONTACTS=1");
    if (key.shared) argv.push_back("-DBEPU_VARIANT_SHARED=1");
    snprintf(define, sizeof(define), "-DBEPU_VARIANT_TYPE_MASK=0x%llxull", key.mask); argv.push_back(define);
    argv.push_back("-DBEPU_UNIT_ENTRY=bepu_special_unit");
    argv.push_back("-I" + unit_source_dir());
    argv.push_back("-x"); argv.push_back("hip");
    argv.push_back(unit_source_dir() + "/bepu_cluster_variant.inc");
    argv.push_back("-o"); argv.push_back(out);
    return argv;
}
// At most BEPUHIP_UNIT_COMPILERS (default 2) compiler processes of this library at a time: a host that creates contexts for many type sets at once (a test suite, a
// fuzzer with BEPUHIP_SPECIALISE=1) must not find dozens of hipcc runs — a gigabyte and a core each — beside its simulation.
// The process is ending (atexit): compiler children are told to stop, workers do not load anything any more and are waited for — a worker that is still inside dlopen or
// the HIP runtime while the runtime's own exit handlers run is a crash at exit.
static std::atomic<bool> g_units_shutdown{false};
static std::mutex g_unit_children_mutex;
static std::vector<pid_t>& unit_children() { static auto* pids = new std::vector<pid_t>(); return *pids; }
struct UnitCompilerSlots {
    std::mutex m; std::condition_variable cv; int running = 0;
    void enter() { std::unique_lock<std::mutex> lock(m); const int most = std::max(1, env_int("BEPUHIP_UNIT_COMPILERS", 2)); cv.wait(lock, [&] { return running < most; }); ++running; }
    void leave() { { std::lock_guard<std::mutex> lock(m); --running; } cv.notify_one(); }
};
static UnitCompilerSlots& unit_compiler_slots() { static auto* slots = new UnitCompilerSlots(); return *slots; }
// Runs the compiler as a child process (stdout / stderr into `log`); true when it exited with 0.
static bool unit_run(const std::vector<std::string>& argv, const std::string& log) {
    struct Slot { Slot() { unit_compiler_slots().enter(); } ~Slot() { unit_compiler_slots().leave(); } } slot;
    std::vector<char*> raw;
    for (auto& a : argv) raw.push_back(const_cast<char*>(a.c_str()));
    raw.push_back(nullptr);
    posix_spawn_file_actions_t actions;
    posix_spawn_file_actions_init(&actions);
    posix_spawn_file_actions_addopen(&actions, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    posix_spawn_file_actions_adddup2(&actions, 1, 2);
    pid_t pid = 0;
    if (g_units_shutdown.load()) { posix_spawn_file_actions_destroy(&actions); return false; }
    const int rc = posix_spawn(&pid, raw[0], &actions, nullptr, raw.data(), environ);
    posix_spawn_file_actions_destroy(&actions);
    if (rc != 0) return false;
    { std::lock_guard<std::mutex> lock(g_unit_children_mutex); unit_children().push_back(pid); }
    if (g_units_shutdown.load()) kill(pid, SIGTERM);
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
    { std::lock_guard<std::mutex> lock(g_unit_children_mutex); auto& pids = unit_children(); pids.erase(std::remove(pids.begin(), pids.end(), pid), pids.end()); }
    return WIFEXITED(status) && WEXITSTATUS(status) == 0 && !g_units_shutdown.load();
}
// The object of `key` as a file: found in one of the cache directories, or compiled into the first writable one. Empty + `why` when neither is possible.
static std::string unit_obtain(const UnitKey& key, std::string& why, bool* compiled_now = nullptr) {
    if (compiled_now) *compiled_now = false;
    if (unit_source_dir().empty() || unit_sources_hash() == 0) { why = "the kernel sources are not next to the library (BEPUHIP_UNIT_SOURCES)"; return std::string(); }
    std::vector<std::string> read_dirs;
    const std::string write_dir = unit_cache_dir(&read_dirs);
    const std::string name = unit_file_name(key);
    for (auto& d : read_dirs) if (access((d + "/" + name).c_str(), R_OK) == 0) return d + "/" + name;
    if (unit_compiler().empty()) { why = "no hipcc on this host (BEPUHIP_HIPCC) and no prebuilt object in the unit cache"; return std::string(); }
    if (write_dir.empty()) { why = "no writable unit cache directory (BEPUHIP_UNIT_CACHE)"; return std::string(); }
    const std::string out = write_dir + "/" + name, tmp = out + ".tmp" + std::to_string((long)getpid()) + "_" + std::to_string((unsigned long)(uintptr_t)&why % 100000);
    if (!unit_run(unit_compile_command(key, tmp), out + ".log")) { unlink(tmp.c_str()); why = "hipcc failed: see " + out + ".log"; return std::string(); }
    if (rename(tmp.c_str(), out.c_str()) != 0) { unlink(tmp.c_str()); why = "could not move the object into the cache"; return std::string(); }
    if (compiled_now) *compiled_now = true;
    return out;
}

static std::mutex g_units_mutex;
static std::map<std::string, SpecialUnit*>& unit_registry() { static auto* registry = new std::map<std::string, SpecialUnit*>(); return *registry; }  // never destroyed: units outlive main's statics

// The unit of `key`, requested if this is the first time anybody asks: the worker finds or compiles the object, loads it, and makes the runtime load its code on `device`.
static SpecialUnit* unit_request(const UnitKey& key, int device, size_t lds_bytes) {
    std::lock_guard<std::mutex> lock(g_units_mutex);
    const std::string name = unit_file_name(key) + "@" + std::to_string(device);
    auto found = unit_registry().find(name);
    if (found != unit_registry().end()) return found->second;
    static const bool at_exit_registered = [] {
        atexit([] {
            g_units_shutdown.store(true);
            { std::lock_guard<std::mutex> lock(g_unit_children_mutex); for (pid_t pid : unit_children()) kill(pid, SIGTERM); }
            std::vector<SpecialUnit*> units;
            { std::lock_guard<std::mutex> lock(g_units_mutex); for (auto& kv : unit_registry()) units.push_back(kv.second); }
            for (SpecialUnit* u : units) { std::lock_guard<std::mutex> lock(u->join_mutex); if (u->worker.joinable()) u->worker.join(); }
        });
        return true;
    }();
    (void)at_exit_registered;
    SpecialUnit* unit = new SpecialUnit();
    unit_registry()[name] = unit;
    unit->worker = std::thread([unit, key, device, lds_bytes] {
        const auto t0 = std::chrono::steady_clock::now();
        std::string why;
        const std::string path = unit_obtain(key, why, &unit->compiled_now);
        int state = kUnitFailed;
        if (g_units_shutdown.load()) { state = kUnitUnavailable; why = "the process is ending"; }
        else if (path.empty()) state = (why.find("hipcc failed") != std::string::npos) ? kUnitFailed : kUnitUnavailable;
        else if (void* dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) {
            typedef const void* (*Entry)(bool);
            if (Entry entry = (Entry)dlsym(dl, "bepu_special_unit")) {
                unit->kernel = entry(false); unit->traced = entry(true);
                // (the first use of a kernel makes the runtime load the object's code onto the device: here, not in front of a frame's launch)
                if (hipSetDevice(device) == hipSuccess && hipFuncSetAttribute(unit->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess &&
                    hipFuncSetAttribute(unit->traced, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess) state = kUnitLoaded;
                else why = "the runtime did not take the object's kernels";
            } else why = "the object has no bepu_special_unit";
        } else why = std::string("dlopen: ") + dlerror();
        unit->path = path; unit->why = why;
        unit->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (env_int("BEPUHIP_PLAN_STATS", 0)) fprintf(stderr, "bepuhip unit %s: %s in %.1f s%s%s\n", unit_file_name(key).c_str(), state == kUnitLoaded ? (unit->compiled_now ? "compiled and loaded" : "loaded from the cache") : "not available",
                                                      unit->seconds, why.empty() ? "" : ": ", why.c_str());
        unit->state.store(state, std::memory_order_release);
    });
    return unit;
}
static void unit_wait(SpecialUnit* unit) {
    std::lock_guard<std::mutex> lock(unit->join_mutex);
    if (unit->worker.joinable()) unit->worker.join();
}
