// beam_replicas_main.cpp -- starts one `beam` process per GPU of a node and waits for them.
//
//   beam_replicas -n N [beam options ...]        the reference's own deployment: N independent replicas, replica i runs
//                                                `beam -g i -D i ...` on GPU i with its own sub-band and its own input
//                                                stream, no communication (README.md:168, src/beamformer.cu:92-100,233)
//   beam_replicas -n N -S [beam options ...]     ONE sub-band sharded over N GPUs: shard i runs `beam -R N -r i -D i -I <id>
//                                                ...`; the detected powers are gathered to shard 0 over RCCL / xGMI
//                                                (include/dsabf.h "Multi-GPU")
// Everything after the launcher's own options is passed to every child unchanged ("{i}" inside an argument is replaced by
// the child's index, e.g. -w detected_{i}.bin or -k ring{i}).  Exit status: the largest child status.
// The launcher itself never touches a GPU (children are exec'ed from a process that has not initialised HIP).
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static std::string self_dir(const char* argv0)
{
    char buf[4096];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    std::string p = n > 0 ? std::string(buf, (size_t)n) : std::string(argv0);
    const size_t slash = p.rfind('/');
    return slash == std::string::npos ? "." : p.substr(0, slash);
}

int main(int argc, char* argv[])
{
    int n = 0, first = 1;
    bool sharded = false;
    std::string beam = self_dir(argv[0]) + "/beam";
    for (; first < argc; first++) {
        if (!strcmp(argv[first], "-n") && first + 1 < argc) n = atoi(argv[++first]);
        else if (!strcmp(argv[first], "-S")) sharded = true;
        else if (!strcmp(argv[first], "-b") && first + 1 < argc) beam = argv[++first];
        else if (!strcmp(argv[first], "--")) { first++; break; }
        else break;
    }
    if (n < 1) {
        fprintf(stderr, "usage: beam_replicas -n N [-S] [-b path/to/beam] [--] [beam options; {i} = replica index]\n");
        return 2;
    }
    // the RCCL id travels through a file in a directory only this launcher's user can enter (mkdtemp: mode 0700, fresh name)
    char id_dir[] = "/tmp/dsabf_comm_XXXXXX";
    if (sharded && !mkdtemp(id_dir)) {
        perror("beam_replicas: mkdtemp");
        return 2;
    }
    const std::string id_file_s = std::string(id_dir) + "/id";
    const char* id_file = id_file_s.c_str();
    std::vector<pid_t> kids;
    for (int i = 0; i < n; i++) {
        std::vector<std::string> args{beam};
        const std::string idx = std::to_string(i);
        if (sharded) {
            args.insert(args.end(), {"-R", std::to_string(n), "-r", idx, "-D", idx, "-I", id_file});
        } else {
            args.insert(args.end(), {"-g", idx, "-D", idx});
        }
        for (int a = first; a < argc; a++) {
            std::string s = argv[a];
            for (size_t pos; (pos = s.find("{i}")) != std::string::npos;) s.replace(pos, 3, idx);
            args.push_back(s);
        }
        std::vector<char*> cargv;
        for (auto& s : args) cargv.push_back(const_cast<char*>(s.c_str()));
        cargv.push_back(nullptr);
        const pid_t pid = fork();
        if (pid < 0) {
            perror("beam_replicas: fork");
            break;
        }
        if (pid == 0) {
            execv(beam.c_str(), cargv.data());
            perror("beam_replicas: exec beam");
            _exit(127);
        }
        kids.push_back(pid);
    }
    int worst = (int)kids.size() == n ? 0 : 1;
    for (size_t i = 0; i < kids.size(); i++) {
        int st = 0;
        if (waitpid(kids[i], &st, 0) < 0) st = 1 << 8;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        if (code) fprintf(stderr, "beam_replicas: replica %zu exited with %d\n", i, code);
        if (code > worst) worst = code;
    }
    if (sharded) {
        unlink(id_file);
        unlink((id_file_s + ".tmp").c_str());
        rmdir(id_dir);
    }
    return worst;
}
