// bf::format_fixed9 (better_flow/event_file.h) against snprintf("%.9f"): special values (signed zeros, exact ties such
// as 1/1024, subnormals, the 2^63 hand-over to snprintf) and three million random doubles of several distributions.
#include <better_flow/common.h>
#include <better_flow/event_file.h>
#include <fstream>
#include <iterator>
#include <random>
int main(){
  std::mt19937_64 rng(7);
  char a[512], b[512];
  long bad=0, n=0;
  auto chk=[&](double x){ int la=bf::format_fixed9(x,a); a[la]=0; std::snprintf(b,sizeof b,"%.9f",x); ++n; if(strcmp(a,b)){ if(bad++<10) printf("MISMATCH %a: %s vs %s\n",x,a,b);} };
  double specials[]={0.0,-0.0,1.0/1024,3.0/1024,5.0/2048,0.5,1.5e-9,2.5e-9,0.5e-9,1e-10,4.9e-324,1e-300,123456789.123456789,9007199254740992.0,9.2e18,1e19,-1e19,1e300,299.99999999950,-150.0000000005, 0.9999999995, 0.99999999949999, 1e15+0.5};
  for(double x:specials){chk(x);chk(-x);}
  for(long i=0;i<3000000;i++){ uint64_t r=rng(); double x; 
    switch(i%5){case 0: x=(double)(int64_t)r*1e-9; break; case 1: x=std::ldexp((double)(r>>11),-(int)(rng()%120)); break; case 2: x=(double)(r%2000000000)/1e6-1000; break; case 3: {uint64_t bits=r; memcpy(&x,&bits,8); if(x!=x|| std::isinf(x)) x=1; if (std::fabs(x)>1e30) x=std::fmod(x,1e6);} break; default: x=((double)(r%4096))/1024.0/ (double)(1<<(rng()%20)); }
    chk(x);} 
  printf("%ld values, %ld mismatches\n",n,bad);
  // bf::write_flow_text: the whole table against snprintf line by line, with thread counts that make it one wave of
  // chunks, several waves (a wave's mapping then starts inside a page) and more threads than chunks
  {
    const size_t N = 131072 * 5 + 777;
    std::vector<uint64_t> ts(N); std::vector<uint16_t> row(N), col(N); std::vector<double> u(N), v(N);
    for (size_t i = 0; i < N; ++i) { ts[i] = 1000000000ull + i * 977ull; row[i] = (uint16_t)(rng() % 260); col[i] = (uint16_t)(rng() % 346);
      u[i] = ((double)(int64_t)(rng() % 2000001) - 1000000.0) / 3000.0; v[i] = std::ldexp((double)(rng() >> 20), -30) - 4000.0; }
    std::string want; want.reserve(N * 48);
    for (size_t i = 0; i < N; ++i) { char line[256]; int l = std::snprintf(line, sizeof line, "%.9f %u %u 1 %.9f %.9f\n", double(ts[i]) / 1000000000, (unsigned)col[i], (unsigned)row[i], v[i], u[i]); want.append(line, (size_t)l); }
    for (int threads : {1, 3, 64}) {
      const char *fn = "/tmp/bf_test_format_out.txt";
      if (!bf::write_flow_text(fn, ts, row, col, u, v, threads)) { printf("write_flow_text failed (%d threads)\n", threads); ++bad; continue; }
      std::ifstream f(fn, std::ios::binary); std::string got((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      if (got != want) { printf("write_flow_text differs with %d threads: %zu vs %zu bytes\n", threads, got.size(), want.size()); ++bad; }
      else printf("write_flow_text, %d threads: %zu bytes equal\n", threads, got.size());
      std::remove(fn);
    }
  }
  return bad!=0; }
