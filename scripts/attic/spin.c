#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
static void *work(void *a){ volatile uint64_t x=1; for(uint64_t i=0;i<(uint64_t)a;i++) x=x*6364136223846793005ULL+1442695040888963407ULL; return (void*)x; }
int main(int c,char**v){ int n=atoi(v[1]); uint64_t it=800000000ULL; pthread_t t[1024]; struct timespec a,b; clock_gettime(CLOCK_MONOTONIC,&a);
 for(int i=0;i<n;i++) pthread_create(&t[i],0,work,(void*)it); for(int i=0;i<n;i++) pthread_join(t[i],0); clock_gettime(CLOCK_MONOTONIC,&b);
 printf("threads %d: %.3fs\n", n, (b.tv_sec-a.tv_sec)+(b.tv_nsec-a.tv_nsec)*1e-9); return 0; }
