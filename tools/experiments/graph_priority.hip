// tools/experiments/graph_priority.hip -- does this HIP runtime take a launch priority on a graph's kernel node?
// Build: hipcc -O2 --offload-arch=gfx950 -w -o tools/experiments/build/graph_priority tools/experiments/graph_priority.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main()
{
    int least = 0, greatest = 0;
    printf("range: %s", hipGetErrorString(hipDeviceGetStreamPriorityRange(&least, &greatest)));
    printf(" least %d greatest %d\n", least, greatest);
    int* d = nullptr;
    hipMalloc((void**)&d, 4);
    hipStream_t s;
    hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
    hipGraph_t g;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
    hipStreamEndCapture(s, &g);
    size_t n = 0;
    hipGraphGetNodes(g, nullptr, &n);
    hipGraphNode_t nodes[4];
    hipGraphGetNodes(g, nodes, &n);
    printf("%zu nodes\n", n);
    hipKernelNodeAttrValue v{};
    printf("get priority: %s", hipGetErrorString(hipGraphKernelNodeGetAttribute(nodes[0], hipKernelNodeAttributePriority, &v)));
    printf(" -> %d\n", v.priority);
    for (int p = -3; p <= 3; ++p) {
        hipKernelNodeAttrValue w{};
        w.priority = p;
        printf("set priority %d: %s\n", p, hipGetErrorString(hipGraphKernelNodeSetAttribute(nodes[0], hipKernelNodeAttributePriority, &w)));
    }
    hipGraphExec_t e;
    printf("instantiate: %s\n", hipGetErrorString(hipGraphInstantiate(&e, g, nullptr, nullptr, 0)));
    printf("launch: %s\n", hipGetErrorString(hipGraphLaunch(e, s)));
    printf("sync: %s\n", hipGetErrorString(hipStreamSynchronize(s)));
    return 0;
}
