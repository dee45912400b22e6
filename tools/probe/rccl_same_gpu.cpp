// Probe: can two processes on ONE GPU form an RCCL communicator (needed to test the C-level multi-GPU entry on a 1-GPU box)?
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

int main(int argc, char** argv)
{
    const int world = argc > 1 ? atoi(argv[1]) : 2;
    // the parent never touches the GPU: rank 0 makes the id and hands it to the others through pipes
    int fds[16][2];
    for (int rank = 1; rank < world; rank++) pipe(fds[rank]);
    for (int rank = 0; rank < world; rank++) {
        pid_t pid = fork();
        if (pid == 0) {
            ncclUniqueId id;
            if (rank == 0) {
                if (ncclGetUniqueId(&id) != ncclSuccess) { printf("ncclGetUniqueId failed\n"); _exit(3); }
                for (int other = 1; other < world; other++) write(fds[other][1], &id, sizeof(id));
            } else {
                if (read(fds[rank][0], &id, sizeof(id)) != sizeof(id)) _exit(4);
            }
            hipSetDevice(0);
            ncclComm_t comm;
            ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
            printf("rank %d: ncclCommInitRank -> %s\n", rank, ncclGetErrorString(r));
            if (r != ncclSuccess) _exit(2);
            int* d = nullptr;
            hipMalloc(&d, 4 * world);
            int v = 100 + rank;
            hipMemcpy(d + rank, &v, 4, hipMemcpyHostToDevice);
            r = ncclAllGather(d + rank, d, 1, ncclInt32, comm, nullptr);
            hipStreamSynchronize(nullptr);
            int out[16];
            hipMemcpy(out, d, 4 * world, hipMemcpyDeviceToHost);
            printf("rank %d: allgather -> %s: %d %d\n", rank, ncclGetErrorString(r), out[0], out[world - 1]);
            ncclCommDestroy(comm);
            _exit(0);
        }
    }
    int bad = 0;
    for (int rank = 0; rank < world; rank++) {
        int status = 0;
        wait(&status);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) bad++;
    }
    printf("children failed: %d\n", bad);
    return bad ? 1 : 0;
}
